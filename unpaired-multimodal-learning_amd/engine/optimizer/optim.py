"""``build_optimizer`` with the reference's signature and semantics
(engine/optimizer/optim.py:15-71), returning an optimizer whose update runs in
the HIP kernels.

The returned object keeps torch.optim's surface that ``finetune.train`` touches
(``zero_grad``, ``step``, ``param_groups``, ``state``), and additionally lets a
``HeadEngine`` adopt its state tensors so the fused step (forward + CE + dW +
update in one launch sequence) updates exactly the state ``step()`` would.
"""
from __future__ import annotations

import torch

AVAI_OPTIMS = ["adam", "sgd", "adamw"]
ADAM_BETAS = (0.9, 0.999)      # Adam / AdamW
MOMENTUM = 0.9                 # SGD
SGD_NESTEROV = False


class HeadOptimizer:
    """State holder for SGD(momentum 0.9) / Adam / AdamW over the head parameters.

    ``state[p]`` holds ``exp_avg``/``exp_avg_sq`` (adam, adamw) or
    ``momentum_buffer`` (sgd) exactly like torch.optim; ``step_count`` is the
    number of optimizer steps taken (Adam bias correction)."""

    def __init__(self, params, name, lr, weight_decay, betas=ADAM_BETAS, momentum=MOMENTUM, eps=1e-8):
        params = list(params)
        if len(params) > 0 and isinstance(params[0], dict):
            if len(params) != 1:
                raise ValueError("HeadOptimizer supports a single parameter group (finetune.py:366)")
            params = list(params[0]["params"])
        self.name = name
        self.defaults = dict(lr=lr, weight_decay=weight_decay, betas=betas, momentum=momentum, eps=eps)
        self.param_groups = [dict(params=params, lr=lr, initial_lr=lr, weight_decay=weight_decay, betas=betas,
                                  momentum=momentum, eps=eps)]
        self.state = {}
        self.step_count = 0
        self._engine = None          # set by the model when the fused path adopts this optimizer

    # -- torch.optim surface ------------------------------------------------------
    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def zero_grad(self, set_to_none=True):
        for p in self.param_groups[0]["params"]:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def state_for(self, p):
        st = self.state.get(p)
        if st is None:
            st = {"exp_avg": torch.zeros_like(p.data), "exp_avg_sq": torch.zeros_like(p.data)}
            if self.name == "sgd":
                st = {"momentum_buffer": torch.zeros_like(p.data)}
            self.state[p] = st
        return st

    def step(self):
        """Unfused update from ``p.grad`` for callers that ran their own backward; the
        arithmetic is the same HIP update kernel the fused step uses."""
        import umlh
        g = self.param_groups[0]
        self.step_count += 1
        live = [p for p in g["params"] if p.grad is not None]
        if not live:
            return
        packed = [p for p in live if not p.data.is_contiguous()]
        if packed:
            # bias=True heads (head.py:65,68): weight and bias are strided VIEWS of one packed [out, in_aug] tensor whose
            # moments live packed in the fused engine; an elementwise update of the views would use detached moments
            raise umlh.UmlhError("HeadOptimizer.step(): %d parameter(s) are views of a packed [weight | bias] tensor (bias=True "
                                 "head); step them through the fused engine (model.fused_engine(optimizer, ...).train_step / "
                                 "finetune.train), which updates the packed tensor and its packed moments" % len(packed))
        ms, vs = [], []
        for p in live:
            st = self.state_for(p)
            ms.append(st["momentum_buffer"] if self.name == "sgd" else st["exp_avg"])
            vs.append(None if self.name == "sgd" else st["exp_avg_sq"])
        # every parameter tensor in one multi-tensor launch (umlh_optimizer_step_multi): the MultiBench model has 70
        umlh.optimizer_step_multi(self.name, [p.data for p in live], [p.grad for p in live], ms, vs, lr=g["lr"],
                                  step=self.step_count, weight_decay=g["weight_decay"], betas=g["betas"], eps=g["eps"],
                                  momentum=g["momentum"])

    def state_dict(self):
        params = self.param_groups[0]["params"]
        return {"name": self.name, "step_count": self.step_count,
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}],
                "state": {i: {k: t.clone() for k, t in self.state[p].items()} for i, p in enumerate(params)
                          if p in self.state}}

    def load_state_dict(self, sd):
        params = self.param_groups[0]["params"]
        self.step_count = sd["step_count"]
        self.param_groups[0].update(sd["param_groups"][0])
        for i, st in sd["state"].items():
            cur = self.state_for(params[int(i)])
            for k, t in st.items():
                cur[k].copy_(t)


def build_optimizer(params_groups, name, lr, weight_decay):
    """Same contract as the reference factory: ``name`` in {"sgd","adam","adamw"},
    AssertionError otherwise (optim.py:22)."""
    assert name in AVAI_OPTIMS, f"Optimizer {name} not found; available optimizers = {AVAI_OPTIMS}"
    if name == "sgd":
        return build_sgd_optimizer(params_groups, lr, weight_decay)
    if name == "adam":
        return build_adam_optimizer(params_groups, lr, weight_decay, betas=ADAM_BETAS)
    return build_adamw_optimizer(params_groups, lr, weight_decay, betas=ADAM_BETAS)


def build_sgd_optimizer(params_groups, lr, weight_decay, momentum=MOMENTUM, nesterov=SGD_NESTEROV):
    if nesterov:
        raise NotImplementedError("nesterov momentum is not used by the reference (optim.py:13)")
    return HeadOptimizer(params_groups, "sgd", lr, weight_decay, momentum=momentum)


def build_adam_optimizer(params_groups, lr, weight_decay, betas=ADAM_BETAS):
    return HeadOptimizer(params_groups, "adam", lr, weight_decay, betas=betas)


def build_adamw_optimizer(params_groups, lr, weight_decay, betas=ADAM_BETAS):
    return HeadOptimizer(params_groups, "adamw", lr, weight_decay, betas=betas)
