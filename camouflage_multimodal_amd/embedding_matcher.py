"""Data glue either side of the fusion path: pairs per-image Region-Graph embeddings with the
Knowledge-Graph category embeddings (the reference's models/multimodal/embedding_matcher.py),
plus a device-resident, pre-packed dataset so a training epoch issues no per-sample host->device
copies (the reference moves every sample at every step, train_multimodal.py:246-251).

On-disk formats kept (reference file:line):
  * RG dict  {image_name: {'node_embeddings' [Nr,128], 'graph_embedding' [1,128], 'num_nodes'}}
    (models/region_graph/extract_rg_embeddings.py:386-390)
  * KG dict  {category: tensor [1,128]}, dict order = category id
    (models/knowledge_graph/extract_kg_embeddings.py:80,102)
  * matched sample dicts: keys image_name, rg_node_embeddings, rg_graph_embedding, kg_embeddings,
    category_ids, num_rg_nodes, num_kg_categories            (embedding_matcher.py:148-156)

``torch.load`` is called with ``weights_only=True``: both files are plain dicts of tensors.
"""
from __future__ import annotations

import os

import torch


class EmbeddingMatcher:
    def __init__(self, rg_embeddings_path=None, kg_embeddings_path=None, category_mapping=None,
                 rg_embeddings=None, kg_embeddings=None):
        """Paths as in the reference [:21-49]; already-loaded dicts may be passed instead."""
        self.rg_embeddings = rg_embeddings if rg_embeddings is not None else torch.load(rg_embeddings_path, weights_only=True)
        self.kg_embeddings = kg_embeddings if kg_embeddings is not None else torch.load(kg_embeddings_path, weights_only=True)
        self.category_mapping = category_mapping
        self.category_to_id = {cat: idx for idx, cat in enumerate(self.kg_embeddings.keys())}
        self.id_to_category = {idx: cat for cat, idx in self.category_to_id.items()}

    def extract_category_from_filename(self, filename):
        """COD10K-CAM-{cam}-{environment}-{seq}-{organism}-{id}: exact, then substring match of the
        organism against the KG categories; None when nothing matches [:51-79]."""
        parts = os.path.splitext(filename)[0].split("-")
        if len(parts) >= 6:
            organism = parts[5]
            if organism in self.kg_embeddings:
                return organism
            for category in self.kg_embeddings.keys():
                if organism.lower() in category.lower() or category.lower() in organism.lower():
                    return category
        return None

    def get_kg_embedding_for_image(self, image_name, use_all_categories=False):
        """-> (kg_emb, category_ids).  All categories: stack of the [1,128] rows -> [Nk,1,128] (the 4-D
        layout the model's input normalisation collapses); one category: [1,1,128]; no match: the
        mean over categories [1,1,128] with placeholder id 0 [:81-115]."""
        if use_all_categories:
            return torch.stack(list(self.kg_embeddings.values())), list(range(len(self.kg_embeddings)))
        category = (self.category_mapping or {}).get(image_name) or self.extract_category_from_filename(image_name)
        if category and category in self.kg_embeddings:
            return self.kg_embeddings[category].unsqueeze(0), [self.category_to_id[category]]
        return torch.stack(list(self.kg_embeddings.values())).mean(dim=0, keepdim=True), [0]

    def create_matched_dataset(self, use_all_kg_categories=True):
        matched = []
        for image_name, rg in self.rg_embeddings.items():
            kg_emb, category_ids = self.get_kg_embedding_for_image(image_name, use_all_categories=use_all_kg_categories)
            matched.append({
                "image_name": image_name,
                "rg_node_embeddings": rg["node_embeddings"],
                "rg_graph_embedding": rg["graph_embedding"],
                "kg_embeddings": kg_emb,
                "category_ids": category_ids,
                "num_rg_nodes": rg["node_embeddings"].shape[0],
                "num_kg_categories": kg_emb.shape[0],
            })
        return matched

    def save_matched_dataset(self, output_path, use_all_kg_categories=True):
        matched = self.create_matched_dataset(use_all_kg_categories)
        torch.save(matched, output_path)
        return matched


class DeviceResidentDataset:
    """All samples packed once into HBM: one [sum Nr, D] matrix of node rows, per-sample offsets,
    one [N, Nk, D] KG tensor and the label vectors.  The 6000-image COD10K set is 1.5 GB of RG rows:
    0.5 % of one MI355X's 288 GB.  ``batch(indices)`` gathers a minibatch with two device-side index
    ops; the training-time augmentation of the reference (N(0, 0.01^2) noise on both streams with
    probability 0.5, train_multimodal.py:173-175) is applied on the device.

    ``samples``: dicts with rg_node_embeddings / kg_embeddings (matched-dataset keys) or rg_node_emb /
    kg_emb (training-sample keys) and mask_label / edge_label / score_label."""

    def __init__(self, samples, device, augment=False, seed=0):
        g = lambda s, *ks: next(s[k] for k in ks if k in s)
        rgs = [g(s, "rg_node_emb", "rg_node_embeddings").float() for s in samples]
        self.nrs = [int(r.shape[0]) for r in rgs]
        self.rg = torch.cat(rgs, dim=0).to(device)
        self.kg = torch.stack([g(s, "kg_emb", "kg_embeddings").float().reshape(-1, rgs[0].shape[1]) for s in samples]).to(device)
        off = torch.zeros(len(samples) + 1, dtype=torch.int64)
        off[1:] = torch.tensor(self.nrs).cumsum(0)
        self.offsets = off
        self.offsets_dev = off.to(device)
        self.mask_label = torch.tensor([int(s["mask_label"]) for s in samples], dtype=torch.int64, device=device)
        self.edge_label = torch.tensor([float(s["edge_label"]) for s in samples], dtype=torch.float32, device=device)
        self.score_label = torch.tensor([float(s["score_label"]) for s in samples], dtype=torch.float32, device=device)
        self.augment = augment
        self.device = device
        self._gen = torch.Generator(device=device if str(device).startswith("cuda") else "cpu")
        self._gen.manual_seed(seed)
        self._seed, self._calls = int(seed), 0

    def __len__(self):
        return len(self.nrs)

    def batch(self, indices, idx_dev=None):
        """-> (rg_packed, nrs, kg, mask_label, edge_label, score_label) for NativeTrainer.step.  On a HIP device the whole
        gather -- packed rows, KG rows, labels, the minibatch's packed offsets and the augmentation -- is ONE launch
        (``camo_gather_batch``); the only host->device traffic is the index vector, and not even that when the caller hands
        the indices as a device tensor too (``idx_dev``: e.g. a slice of an epoch's draw copied over once).  ``nrs`` carries
        the offsets as ``nrs.offsets_dev`` for ``FusionEngine.make_batch``."""
        indices = [int(i) for i in indices]
        nrs = NrList(self.nrs[i] for i in indices)
        T, B = sum(nrs), len(indices)
        idx = idx_dev if idx_dev is not None else torch.tensor(indices, device=self.device)
        if self.rg.is_cuda and B <= 4096:
            from . import _lib
            from .engine import _stream_ptr
            dev = self.rg.device
            rg = torch.empty(T, self.rg.shape[1], dtype=torch.float32, device=dev)
            kg = torch.empty(B, self.kg.shape[1], self.kg.shape[2], dtype=torch.float32, device=dev)
            offs = torch.empty(B + 1, dtype=torch.int32, device=dev)
            y = torch.empty(B, dtype=torch.int64, device=dev); e = torch.empty(B, dtype=torch.float32, device=dev); s = torch.empty_like(e)
            self._calls += 1
            with (torch.cuda.device(dev) if torch.cuda.current_device() != dev.index else _NULLCTX):
                _lib.check(_lib.lib().camo_gather_batch(self.rg.data_ptr(), self.offsets_dev.data_ptr(), self.kg.data_ptr(), self.mask_label.data_ptr(),
                                                        self.edge_label.data_ptr(), self.score_label.data_ptr(), idx.data_ptr(), B, T, self.rg.shape[1],
                                                        self.kg.shape[1] * self.kg.shape[2], rg.data_ptr(), kg.data_ptr(), offs.data_ptr(), y.data_ptr(),
                                                        e.data_ptr(), s.data_ptr(), 0.01 if self.augment else 0.0,
                                                        (self._seed * 0x9E3779B97F4A7C15 + self._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF,
                                                        _stream_ptr(dev)), "camo_gather_batch")
            nrs.offsets_dev = offs
            return rg, nrs, kg, y, e, s
        # host-side datasets (tests on CPU): the same gather with torch index ops
        starts = self.offsets_dev[idx]
        lens = self.offsets_dev[idx + 1] - starts
        offs = torch.zeros(B + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(lens, 0, out=offs[1:])
        # packed row r of the minibatch = dataset row starts[b] + (r - offs[b]) for the sample b it belongs to
        rows = torch.arange(T, device=self.device) + torch.repeat_interleave(starts - offs[:-1], lens, output_size=T)
        nrs.offsets_dev = offs.to(torch.int32)
        rg = self.rg.index_select(0, rows)
        kg = self.kg.index_select(0, idx)
        if self.augment:
            coin = torch.rand(B, generator=self._gen, device=self.device) > 0.5
            per_row = torch.repeat_interleave(coin, lens, output_size=T)
            rg = rg + torch.randn(rg.shape, generator=self._gen, device=self.device) * 0.01 * per_row[:, None]
            kg = kg + torch.randn(kg.shape, generator=self._gen, device=self.device) * 0.01 * coin[:, None, None]
        return rg, nrs, kg, self.mask_label[idx], self.edge_label[idx], self.score_label[idx]


class _NullCtx:
    def __enter__(self): return None
    def __exit__(self, *a): return False


_NULLCTX = _NullCtx()


class NrList(list):
    """The per-sample node counts of a packed minibatch (host ints) plus, as ``offsets_dev``, their packed row offsets as an
    int32 device tensor [B + 1] when the producer already has them there."""
    offsets_dev = None
