"""The shared UML classifier head with the reference's class names, constructor
arguments, attributes and methods (engine/models/head.py:39-141), computing on
the HIP kernels of ``umlh``.

In this build the backbone is the identity over PRE-EXTRACTED feature rows
(what features.py writes; README "linear probe" regime): ``vision_model`` /
``clip_encoder`` name the feature width instead of a network to download.

    UML(vision_model, text_indim, num_classes, bias=False, learnable_temp=False, freeze_backbone=False)
    UMLClip(clip_encoder, num_classes, logit_scale_init=4.60517, bias=False, learnable_temp=False, ...)
    model(images, text_features=None) -> (img_logits, txt_logits | None)
    model.head.weight, model.img_proj, model.num_classes, model.shared_dim,
    model.extract_features, model.extract_raw_features, model.zero_shot_init(text_ds)

``bias=True``: weight and bias of a layer are views of one packed tensor and the
kernels see rows [x | 1 | 0...] (see ``_HeadBase._pack_head``); with img_proj both layers carry their bias and one row of
the packed projection is a frozen constant (``umlh_freeze_proj_row``).

``forward`` returns the logits tensors (computed by ``umlh_logits``); training
goes through ``fused_engine(optimizer)`` -> ``HeadEngine.train_step`` which never
materialises logits (see finetune.train in this package).
"""
from __future__ import annotations

import math

import torch
from torch import nn

# embed_dim of the CLIP checkpoints the reference can name (engine/clip/clip.py:29-36)
CLIP_EMBED_DIM = {"RN50": 1024, "RN101": 512, "RN50x4": 640, "RN50x16": 768, "ViT-B/32": 512, "ViT-B/16": 512,
                  "ViT-L/14": 768}


class FeatureRows(nn.Module):
    """Identity 'backbone': the input already is the [B, num_features] feature matrix."""

    def __init__(self, num_features):
        super().__init__()
        self.num_features = int(num_features)
        self.embed_dim = int(num_features)
        self.num_classes = 0

    def forward(self, x):
        return x

    encode_image = forward


def _feature_width(spec, table=None):
    if isinstance(spec, nn.Module):
        return spec
    if isinstance(spec, int):
        return FeatureRows(spec)
    if isinstance(spec, str):
        if table and spec in table:
            return FeatureRows(table[spec])
        for prefix in ("features:", "identity:"):
            if spec.startswith(prefix):
                return FeatureRows(int(spec[len(prefix):]))
    raise ValueError(f"backbone {spec!r}: this build runs on pre-extracted features; pass the feature width "
                     f"(int or 'features:<d>') or a known CLIP encoder name {sorted(CLIP_EMBED_DIM)}")


def get_text_dataset_per_class(text_dataset):
    """label -> list of (embedding, eot_index) in dataset order (head.py:7-19)."""
    groups = {}
    for item in text_dataset:
        emb, label = item[0], item[1]
        eot = item[2] if len(item) > 2 else None
        groups.setdefault(int(label), []).append((emb, eot))
    return groups


def _text_rows(text_dataset):
    if hasattr(text_dataset, "input_tensor") and isinstance(text_dataset.input_tensor, torch.Tensor):
        return text_dataset.input_tensor, text_dataset.label_tensor
    feats = torch.stack([torch.as_tensor(it[0]) for it in text_dataset])
    labels = torch.as_tensor([int(it[1]) for it in text_dataset])
    return feats, labels


def get_zero_shot_weights(text_dataset, num_classes, in_features, device="cuda"):
    """Per-class mean text embedding, L2-normalised rows, classes without text stay
    zero (head.py:22-37) -- computed by ``umlh_zero_shot_init`` on ``device``; the
    [num_classes, in_features] weight matrix is returned on the CPU like the reference."""
    import umlh
    dev = torch.device("cuda:0" if device == "cuda" else device)
    eng = umlh.HeadEngine(in_features, in_features, num_classes, optimizer="sgd", max_rows_img=32, max_rows_txt=32,
                          device=dev)
    feats, labels = _text_rows(text_dataset)
    eng.zero_shot_init(feats, labels)
    w = eng.w_head.detach().cpu().clone()
    eng.close()
    return w


class _HeadBase(nn.Module):
    def _init_common(self, backbone, shared_dim, num_classes, bias, scales, learnable):
        self._bias = bool(bias)
        self._packed = None              # bias=True: [C, d_aug] = [weight | bias | 0...], the tensor the kernels own
        self.num_classes = num_classes
        self.vision_model = backbone
        self.shared_dim = shared_dim
        self._learnable_temp = bool(learnable)
        self._engines = {}
        self._scales = torch.tensor(scales, dtype=torch.float32)

    # ---- packed logit scales: the kernels read a device float[2] -------------------
    def _scale_views(self):
        return self._scales[0], self._scales[1]

    def _repack_scales(self, device=None):
        pass

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._scales = fn(self._scales)
        self._after_move()
        self._engines = {}                       # device pointers changed
        return self

    def _after_move(self):
        self._pack_head()

    # ---- head with bias (head.py:65,68 ``bias=True``) -----------------------------------
    # The kernels run a bias-free head over rows [x | 1 | 0...] (HeadEngine ``bias_from``): ``head.weight`` and ``head.bias``
    # are VIEWS of one packed [C, d_aug] tensor (d_aug = d + 1 rounded up to 128: valid for both precision modes), so the
    # reference's parameter names, shapes, initialisation and state_dict keys stay as they are.
    @staticmethod
    def _pack_linear(lin, rows_aug=None):
        """[out, in] weight + [out] bias of ``lin`` -> packed [rows, in_aug] = [weight | bias | 0...] (in_aug = in + 1 rounded up
        to 128; ``rows_aug`` > out adds zero rows); weight and bias become views of it."""
        w, b = lin.weight, lin.bias
        out, d = w.shape
        d_aug = (d + 1 + 127) // 128 * 128
        packed = torch.zeros(rows_aug or out, d_aug, dtype=torch.float32, device=w.device)
        packed[:out, :d] = w.data
        packed[:out, d] = b.data
        w.data = packed[:out, :d]
        b.data = packed[:out, d]
        return packed

    def _pack_head(self):
        if not getattr(self, "_bias", False) or not hasattr(self, "head"):
            return
        self._packed = self._pack_linear(self.head)
        self._packed_proj = None
        if getattr(self, "img_proj", None) is not None:
            # img_proj with bias: its output rows are [h | 1 | 0...] so that the head's bias column sees a 1 on projected image
            # rows too -- row d_sh of the packed projection copies the ones column of the input (frozen: umlh_freeze_proj_row)
            d_sh, d_img = self.img_proj.weight.shape
            self._packed_proj = self._pack_linear(self.img_proj, rows_aug=self._packed.shape[1])
            self._packed_proj[d_sh, d_img] = 1.0

    def _packed_state(self, optimizer, opt_name, lin, packed, tag):
        """Optimizer moments of (weight, bias) of ``lin`` as views of packed tensors shaped like ``packed`` (what the kernels update)."""
        store = optimizer.__dict__.setdefault("_packed_state", {})
        pk = store.get(tag)
        w, b = lin.weight, lin.bias
        out, d = w.shape
        if pk is None or pk[0].shape != packed.shape or pk[0].device != packed.device:
            names = ["momentum_buffer"] if opt_name == "sgd" else ["exp_avg", "exp_avg_sq"]
            pk = tuple(torch.zeros_like(packed) for _ in names)
            for t, nm in zip(pk, names):                     # adopt moments the optimizer may already hold (resumed state)
                for p, view in ((w, t[:out, :d]), (b, t[:out, d])):
                    old = optimizer.state.get(p, {}).get(nm)
                    if old is not None:
                        view.copy_(old)
            optimizer.state[w] = {nm: t[:out, :d] for t, nm in zip(pk, names)}
            optimizer.state[b] = {nm: t[:out, d] for t, nm in zip(pk, names)}
            store[tag] = pk
        return pk

    # ---- fused path ------------------------------------------------------------------
    @property
    def _has_proj(self):
        return getattr(self, "img_proj", None) is not None

    def _is_feature_backbone(self):
        return isinstance(self.vision_model, FeatureRows)

    def fused_engine(self, optimizer=None, max_rows_img=4096, max_rows_txt=4096, precision="fp32"):
        """HeadEngine bound to THIS module's parameter storage (and to ``optimizer``'s
        state tensors when given), so the fused kernels update them in place."""
        import umlh
        if not self._is_feature_backbone():
            raise umlh.UmlhError("fused path needs pre-extracted feature rows (FeatureRows backbone)")
        dev = self.head.weight.device
        opt_name = optimizer.name if optimizer is not None else "sgd"
        key = (id(optimizer) if optimizer is not None else None, int(max_rows_img), int(max_rows_txt), precision)
        eng = self._engines.get(key)
        if eng is not None:
            return eng
        g = optimizer.param_groups[0] if optimizer is not None else dict(weight_decay=0.0, betas=(0.9, 0.999),
                                                                         eps=1e-8, momentum=0.9)
        if self._bias:
            d_aug = self._packed.shape[1]
            proj = self._packed_proj is not None
            eng = umlh.HeadEngine(self._packed_proj.shape[1] if proj else d_aug, d_aug, self.num_classes, has_proj=proj,
                                  learnable_temp=self._learnable_temp, optimizer=opt_name, weight_decay=g["weight_decay"],
                                  betas=g["betas"], eps=g["eps"], momentum=g["momentum"], max_rows_img=max_rows_img,
                                  max_rows_txt=max_rows_txt, precision=precision, device=dev,
                                  bias_from=(self.img_proj.weight.shape[1], self.shared_dim) if proj else self.shared_dim)
            bind = dict(w_head=self._packed, scales=self._scales)
            if proj:
                bind["w_proj"] = self._packed_proj
            if optimizer is not None:
                pk = self._packed_state(optimizer, opt_name, self.head, self._packed, "head")
                bind["m_head"] = pk[0]
                if len(pk) > 1:
                    bind["v_head"] = pk[1]
                if proj:
                    pk = self._packed_state(optimizer, opt_name, self.img_proj, self._packed_proj, "proj")
                    bind["m_proj"] = pk[0]
                    if len(pk) > 1:
                        bind["v_proj"] = pk[1]
                if self._learnable_temp:
                    self._bind_scale_state(optimizer, opt_name, dev, bind)
            eng.rebind(**bind)
            self._engines[key] = eng
            return eng
        eng = umlh.HeadEngine(self.vision_model.num_features, self.shared_dim, self.num_classes,
                              has_proj=self._has_proj, learnable_temp=self._learnable_temp, optimizer=opt_name,
                              weight_decay=g["weight_decay"], betas=g["betas"], eps=g["eps"], momentum=g["momentum"],
                              max_rows_img=max_rows_img, max_rows_txt=max_rows_txt, precision=precision, device=dev)
        bind = dict(w_head=self.head.weight.data, scales=self._scales)
        if self._has_proj:
            bind["w_proj"] = self.img_proj.weight.data
        if optimizer is not None:
            def mv(p):
                st = optimizer.state_for(p)
                return (st["momentum_buffer"], None) if opt_name == "sgd" else (st["exp_avg"], st["exp_avg_sq"])
            bind["m_head"], v = mv(self.head.weight)
            if v is not None:
                bind["v_head"] = v
            if self._has_proj:
                bind["m_proj"], v = mv(self.img_proj.weight)
                if v is not None:
                    bind["v_proj"] = v
            if self._learnable_temp:
                self._bind_scale_state(optimizer, opt_name, dev, bind)
        eng.rebind(**bind)
        self._engines[key] = eng
        return eng

    def _bind_scale_state(self, optimizer, opt_name, dev, bind):
        """Learnable logit scales (head.py:69-70): the kernels keep the two scales' moments packed as float[2]; expose them to
        the optimizer as per-parameter views so state_dict()/step() see the same storage."""
        pk = getattr(optimizer, "_packed_scale_state", None)
        if pk is None:
            pk = (torch.zeros(2, device=dev), torch.zeros(2, device=dev))
            optimizer._packed_scale_state = pk
            for j, p in enumerate((self.img_scale, self.txt_scale)):
                optimizer.state[p] = ({"momentum_buffer": pk[0][j]} if opt_name == "sgd" else
                                      {"exp_avg": pk[0][j], "exp_avg_sq": pk[1][j]})
        bind["m_scales"], bind["v_scales"] = pk

    def _infer_engine(self, rows):
        for eng in self._engines.values():     # logits / eval always run the exact fp32 kernels
            if eng.precision == "fp32" and min(eng.cfg.max_rows_img, eng.cfg.max_rows_txt) >= rows:
                return eng
        cap = 256
        while cap < rows:
            cap *= 2
        return self.fused_engine(None, cap, cap)

    def _logits(self, feats, modality):
        import umlh
        feats = feats.to(torch.float32).contiguous()
        dummy = torch.zeros(feats.shape[0], dtype=torch.int64, device=feats.device)
        return self._infer_engine(feats.shape[0]).logits(umlh.RowBatch(feats, dummy), modality)

    # ---- reference surface ------------------------------------------------------------
    def forward(self, images, text_features=None):
        feats = self.vision_model(images)
        img_logits = self._logits(feats, 0)
        if text_features is not None:
            return img_logits, self._logits(text_features, 1)
        return img_logits, None

    def extract_raw_features(self, images):
        return self.vision_model(images)

    def zero_shot_init(self, zeroshot_dataset):
        print("=> Initializing head with zero-shot weights")
        dev = self.head.weight.device
        # a sweep initialises every grid point from the same text rows: the weights are computed once per dataset object
        key = (id(zeroshot_dataset), self.num_classes, self.shared_dim)
        cache = getattr(zeroshot_dataset, "_umlh_zero_shot", None)
        if cache is not None and cache[0] == key:
            w = cache[1]
        else:
            w = get_zero_shot_weights(zeroshot_dataset, self.num_classes, self.shared_dim,
                                      device=dev if dev.type == "cuda" else "cuda")
            try:
                zeroshot_dataset._umlh_zero_shot = (key, w)
            except AttributeError:
                pass
        # in-place: keeps engine bindings valid (the reference rebinds .data, head.py:98)
        self.head.weight.data.copy_(w.to(dev))


class UML(_HeadBase):
    def __init__(self, vision_model, text_indim, num_classes, bias=False, learnable_temp=False,
                 freeze_backbone=False):
        super().__init__()
        backbone = _feature_width(vision_model)
        self.img_proj = None
        shared = backbone.num_features
        self._init_common(backbone, shared, num_classes, bias, [1.0, 1.0], learnable_temp)
        # parameter creation order == reference (img_proj, head, img_scale, txt_scale): the
        # nn.Linear initialisers consume the global RNG identically (seed parity)
        if text_indim > 0:
            self.img_proj = nn.Linear(backbone.num_features, text_indim, bias=bool(bias))
            self.shared_dim = text_indim
        self.head = nn.Linear(self.shared_dim, num_classes, bias=bool(bias))
        if learnable_temp:
            self.img_scale = nn.Parameter(torch.tensor(1.0))
            self.txt_scale = nn.Parameter(torch.tensor(1.0))
        else:
            self.img_scale = torch.tensor(1.0)
            self.txt_scale = torch.tensor(1.0)
        self._after_move()

    def _after_move(self):
        # keep img_scale / txt_scale as views of the packed device float[2]
        if not hasattr(self, "head"):
            return
        self._pack_head()
        dev = self.head.weight.device
        vals = torch.stack([self.img_scale.detach().to(dev, torch.float32).reshape(()),
                            self.txt_scale.detach().to(dev, torch.float32).reshape(())])
        self._scales = vals.contiguous()
        if self._learnable_temp:
            self.img_scale.data = self._scales[0]
            self.txt_scale.data = self._scales[1]
        else:
            self.img_scale, self.txt_scale = self._scales[0], self._scales[1]

    def extract_features(self, images):
        feats = self.vision_model(images)
        if self.img_proj is None:
            return feats
        import umlh
        feats = feats.to(torch.float32).contiguous()
        dummy = torch.zeros(feats.shape[0], dtype=torch.int64, device=feats.device)
        h = self._infer_engine(feats.shape[0]).project(umlh.RowBatch(feats, dummy))
        return h[:, :self.shared_dim].contiguous() if self._bias else h      # (the packed projection appends [1 | 0...])


class UMLClip(_HeadBase):
    def __init__(self, clip_encoder, num_classes, logit_scale_init=math.log(1 / 0.07), bias=False,
                 learnable_temp=False, freeze_backbone=False):
        super().__init__()
        backbone = _feature_width(clip_encoder, CLIP_EMBED_DIM)
        self.img_proj = None
        s = math.exp(float(logit_scale_init))
        self._init_common(backbone, backbone.embed_dim, num_classes, bias, [s, s], False)
        self.head = nn.Linear(self.shared_dim, num_classes, bias=bool(bias))
        self.logit_scale = torch.tensor(float(logit_scale_init))      # fixed scale (head.py:125)
        self._pack_head()

    def extract_features(self, images):
        """Missing in the reference class although train() calls it; identity here."""
        return self.vision_model(images)
