"""Hyper-parameter grids of the reference sweeps (engine/optimizer/default.py:1-60),
restated as data: every list value is one axis of the cartesian product that
``finetune.sweep`` walks."""

_COMMON = dict(optim="adamw", lr_scheduler="cosine", max_iter=[12800], warmup_iter=50, warmup_type="linear",
               warmup_min_lr=1e-5, dropout=[0.0])


def _grid(**kw):
    g = dict(_COMMON)
    g.update(kw)
    # key order matters: it is the order of the product and of hparam_str
    order = ["optim", "lr", "weight_decay", "lr_scheduler", "batch_size", "max_iter", "warmup_iter",
             "warmup_type", "warmup_min_lr", "dropout", "learnable_temp", "patience"]
    return {k: g[k] for k in order}


HYPER_DICT = {
    # full fine-tuning
    "full_ds_full_model_finetune": _grid(lr=[5e-05], weight_decay=[0.0, 0.01, 0.001], batch_size=[64],
                                         learnable_temp=[False], patience=[10]),
    # linear probe on CLIP features
    "clip_linear": _grid(lr=[0.001, 0.0001], weight_decay=[0.0, 0.01, 0.001], batch_size=[32],
                         learnable_temp=[False], patience=[5]),
    # linear probe on unimodal vision + language encoders
    "linear": _grid(lr=[0.001, 0.0001], weight_decay=[0.0, 0.01, 0.001], batch_size=[8, 32],
                    learnable_temp=[True], patience=[10]),
    "audio": _grid(lr=[0.1, 0.01, 0.001, 0.0001], weight_decay=[0.0, 0.01, 0.0001], batch_size=[8],
                   learnable_temp=[False], patience=[5]),
}
