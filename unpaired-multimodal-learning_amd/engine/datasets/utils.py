"""Feature-row datasets and loaders for the head fine-tune loop.

``TextTensorDataset`` keeps the reference's constructor and reductions
(engine/datasets/utils.py:48-107).  ``FeatureLoader`` stands where
``DataLoader(DatasetWrapper(...))`` / ``DataLoader(text_ds)`` stand in
``finetune.setup`` (finetune.py:370-383): it yields the same batch containers
(dict with 'img'/'label' for images, 3-tuple for text) in the same seed-identical
order, but the rows live in HBM and a batch is an int64 index vector, not a
collated copy -- the fused step gathers rows inside the GEMM operand load.
"""
from __future__ import annotations

import torch


class TextTensorDataset(torch.utils.data.Dataset):
    """(features [N,d], labels [N], eot_indices [N]) with the reference's ``n_shots``
    options: int -> per-class random subsample, "average" -> one mean row per class."""

    def __init__(self, input_tensor, label_tensor, eot_indices, n_shots=None):
        self.input_tensor, self.label_tensor, self.eot_indices = input_tensor, label_tensor, eot_indices
        if isinstance(n_shots, int):
            keep = self._select_n_shots(label_tensor, n_shots)
            if isinstance(input_tensor, list):
                self.input_tensor = [input_tensor[i] for i in keep.tolist()]
            else:
                self.input_tensor = input_tensor[keep]
            self.label_tensor, self.eot_indices = label_tensor[keep], eot_indices[keep]
            print(f"=> Using {n_shots} text shots per class, with total of {len(self)} samples")
        elif isinstance(n_shots, str) and n_shots.lower() == "average":
            self.input_tensor, self.label_tensor, self.eot_indices = self._average_features(
                input_tensor, label_tensor, eot_indices)
            print(f"=> Averaging text features per class, with total of {len(self)} samples")
        elif n_shots is not None:
            raise ValueError("n_shots must be an int, None, or 'average'")

    @staticmethod
    def _select_n_shots(labels, n_shots):
        # one torch.randperm per class from the GLOBAL generator, classes in sorted
        # order: the draw sequence the reference makes (utils.py:76-86)
        picked = []
        for c in torch.unique(labels):
            rows = torch.nonzero(labels == c, as_tuple=True)[0]
            picked.append(rows[torch.randperm(rows.numel())[:min(n_shots, rows.numel())]])
        return torch.cat(picked)

    @staticmethod
    def _average_features(inputs, labels, eot_indices):
        classes = torch.unique(labels)
        means = torch.stack([inputs[labels == c].mean(dim=0) for c in classes])
        first_eot = [eot_indices[labels == c][0] for c in classes]
        eot = torch.stack(first_eot) if isinstance(first_eot[0], torch.Tensor) else torch.tensor(first_eot)
        return means, classes, eot

    def __getitem__(self, index):
        return self.input_tensor[index], self.label_tensor[index], self.eot_indices[index]

    def __len__(self):
        return self.input_tensor.size(0) if isinstance(self.input_tensor, torch.Tensor) else len(self.input_tensor)


class TensorDataset(torch.utils.data.Dataset):
    def __init__(self, input_tensor, label_tensor):
        self.input_tensor, self.label_tensor = input_tensor, label_tensor

    def __getitem__(self, index):
        return self.input_tensor[index], self.label_tensor[index]

    def __len__(self):
        return self.input_tensor.size(0)


class FeatureTable:
    """Device-resident feature rows + labels (what features.py writes as
    {'features','labels',...}: features.py:180-184,143-149)."""

    def __init__(self, features, labels, device, extra=None):
        self.features = features.to(device=device, dtype=torch.float32).contiguous()
        self.labels = labels.to(device=device, dtype=torch.int64).contiguous()
        self.extra = extra.to(device) if isinstance(extra, torch.Tensor) else extra
        self.device = self.features.device
        self._bf16 = None

    def features_bf16(self):
        """bf16 shadow of the table (built once with umlh_to_bf16) for bf16-mode engines."""
        if self._bf16 is None:
            import umlh
            self._bf16 = umlh.to_bf16(self.features)
        return self._bf16

    def __len__(self):
        return self.features.shape[0]


class FeatureLoader:
    """Iterable over batches of a FeatureTable with torch ``DataLoader`` semantics
    (``batch_size``, ``shuffle``, ``drop_last=False``) and DataLoader's RNG protocol,
    so that under the same global seed the batches are the ones the reference's
    loaders deliver:

      * ``iter(loader)`` draws the iterator's base seed from the global CPU generator;
      * the first ``next`` of a shuffled loader draws the sampler seed the same way
        and permutes with a private generator seeded by it.

    ``iter_index()`` yields device int64 index vectors (zero-copy batches for the
    fused step); plain iteration yields the reference's batch containers."""

    def __init__(self, table: FeatureTable, batch_size, shuffle=False, drop_last=False, kind="image",
                 order_rng="torch-cpu", generator=None):
        assert kind in ("image", "text") and order_rng in ("torch-cpu", "device")
        # generator: private torch.Generator the iterator/sampler seeds are drawn from instead of the
        # global CPU generator (DataLoader's own ``generator=`` argument) -- concurrent sweep workers
        # each own one, so their batch orders do not depend on thread interleaving
        self.generator = generator
        self.table, self.batch_size, self.shuffle, self.drop_last, self.kind = table, int(batch_size), shuffle, drop_last, kind
        self.dataset = table
        # "torch-cpu": the reference's DataLoader order bit for bit (CPU mt19937 randperm + H2D per
        # epoch, ~20 ms for ImageNet); "device": same distribution, permutation drawn on the GPU
        # (no host work on the step path) -- for throughput runs where order parity is not asserted
        self.order_rng = order_rng
        self._dev_gen = None

    def __len__(self):
        n = len(self.table)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def iter_index(self):
        """Iterator over device int64 index vectors.  Like ``iter(DataLoader)`` it draws
        its base seed on creation and the sampler seed at the first ``next``."""
        return _IndexIter(self)

    def pack(self, idx):
        f, y = self.table.features[idx], self.table.labels[idx]
        if self.kind == "image":
            return {"img": f, "label": y, "index": idx}
        eot = self.table.extra[idx] if isinstance(self.table.extra, torch.Tensor) else torch.zeros_like(y)
        return (f, y, eot)

    def __iter__(self):
        for idx in self.iter_index():
            yield self.pack(idx)


class _IndexIter:
    def __init__(self, loader: FeatureLoader):
        self.loader = loader
        torch.empty((), dtype=torch.int64).random_(generator=loader.generator)   # DataLoader iterator base seed
        self.order = None
        self.pos = 0

    def __iter__(self):
        return self

    def __next__(self):
        ld = self.loader
        n, bs = len(ld.table), ld.batch_size
        if self.order is None:
            if ld.shuffle:
                seed = int(torch.empty((), dtype=torch.int64).random_(generator=ld.generator).item())   # RandomSampler seed
                if ld.order_rng == "device":
                    import umlh
                    self.order = umlh.random_permutation(n, seed, ld.table.device)   # one tiny kernel, no sort
                else:
                    g = torch.Generator()
                    g.manual_seed(seed)
                    self.order = torch.randperm(n, generator=g).to(ld.table.device, non_blocking=True)
            else:
                self.order = torch.arange(n, dtype=torch.int64, device=ld.table.device)
        if self.pos >= n or (ld.drop_last and self.pos + bs > n):
            raise StopIteration
        idx = self.order[self.pos:self.pos + bs]
        self.pos += bs
        return idx

    # -- bulk access for the blockwise training loop: many consecutive batches of the CURRENT epoch as one slice --
    def remaining(self):
        """Batches left in the current epoch that ``__next__`` would deliver without raising (0 before the first
        batch of an epoch has been drawn: that call draws the sampler seed and must go through ``__next__``)."""
        if self.order is None:
            return 0
        ld = self.loader
        n, bs = len(ld.table), ld.batch_size
        left = n - self.pos
        return max(0, left // bs if ld.drop_last else (left + bs - 1) // bs)

    def take_span(self, k):
        """The next ``k <= remaining()`` batches as (one index slice, [batch sizes])."""
        ld = self.loader
        n, bs = len(ld.table), ld.batch_size
        end = min(n, self.pos + k * bs)
        idx = self.order[self.pos:end]
        sizes = [bs] * ((end - self.pos) // bs)
        if (end - self.pos) % bs:
            sizes.append((end - self.pos) % bs)
        assert len(sizes) == k
        self.pos = end
        return idx, sizes
