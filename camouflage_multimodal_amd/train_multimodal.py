"""Native trainer: the reference's training schedule (models/multimodal/train_multimodal.py)
with the per-sample Python loop replaced by one packed minibatch per optimizer step.

Reference semantics kept (file:line of the reference):
  * a minibatch is a list of sample dicts (``collate_fn`` is the identity)            [:191-192]
  * gradients are the SUM over the samples of the minibatch, then one
    ``clip_grad_norm_(1.0)`` and one ``AdamW.step``                                     [:238-279]
  * the loss of a sample is 3*focal + CE + 0.5*BCE + 0.3*MSE at batch size 1            [:256-268]
  * reported loss = sum of sample losses / number of samples; F1 from arg-max preds     [:281-301]
  * validation: eval-mode forward, plain cross-entropy on the mask logits, per-class
    accuracy                                                                            [:304-342]
  * best-checkpoint dict format and ``training_history_fixed.json``                     [:422-427,:464-474,:489]

What changes is where the arithmetic runs: ``forward -> loss -> backward -> clip -> AdamW`` are
five enqueue-only C-ABI calls on one HIP stream; predictions and loss terms stay on the device
and are read back once per epoch instead of three ``.item()`` syncs per sample [:269-275].
"""
from __future__ import annotations

import json
import os

import torch

from .fusion_model import build_multimodal_model
from .losses import multitask_loss
from .optim import FusedClipAdamW


def collate_fn(batch):
    return batch


def calculate_f1_score(predictions, labels):
    """Per-class precision/recall/F1 with the reference's 1e-8 smoothing [:197-220]."""
    predictions = torch.as_tensor(predictions); labels = torch.as_tensor(labels)
    tp = ((predictions == 1) & (labels == 1)).sum(); fp = ((predictions == 1) & (labels == 0)).sum()
    fn = ((predictions == 0) & (labels == 1)).sum(); tn = ((predictions == 0) & (labels == 0)).sum()
    precision_1 = tp / (tp + fp + 1e-8); recall_1 = tp / (tp + fn + 1e-8)
    f1_class_1 = 2 * (precision_1 * recall_1) / (precision_1 + recall_1 + 1e-8)
    precision_0 = tn / (tn + fn + 1e-8); recall_0 = tn / (tn + fp + 1e-8)
    f1_class_0 = 2 * (precision_0 * recall_0) / (precision_0 + recall_0 + 1e-8)
    return {"f1_class_0": f1_class_0, "f1_class_1": f1_class_1, "f1_avg": (f1_class_0 + f1_class_1) / 2,
            "precision_1": precision_1, "recall_1": recall_1}


def pack_samples(batch, device):
    """List of reference-style sample dicts -> packed device tensors.
    ``rg_node_emb`` [Nr,128]; ``kg_emb`` [Nk,128] or the [Nk,1,128] layout EmbeddingMatcher
    produces (embedding_matcher.py:95-96)."""
    nrs = [int(s["rg_node_emb"].shape[0]) for s in batch]
    rg = torch.cat([s["rg_node_emb"].reshape(n, -1) for s, n in zip(batch, nrs)], dim=0)
    kg = torch.stack([s["kg_emb"].reshape(-1, s["kg_emb"].shape[-1]) for s in batch])
    y = torch.tensor([int(s["mask_label"]) for s in batch], dtype=torch.int64)
    e = torch.tensor([float(s["edge_label"]) for s in batch], dtype=torch.float32)
    sc = torch.tensor([float(s["score_label"]) for s in batch], dtype=torch.float32)
    to = lambda t: t.to(device, non_blocking=True)
    return to(rg.float()), nrs, to(kg.float()), to(y), to(e), to(sc)


class NativeTrainer:
    """One optimizer step = one packed minibatch through the HIP path."""

    def __init__(self, model, lr=5e-4, weight_decay=1e-4, max_norm=1.0, grad_allreduce=None, keep_grads=False, fused_call=True):
        """``keep_grads``: leave the clipped gradients in ``param.grad`` after the step, as the reference's
        clip_grad_norm_ does (costs one extra clearing pass per step); by default the optimizer kernel
        clears the gradient buffer itself, which is the reference's per-minibatch ``zero_grad()`` [:239]
        moved to the end of the previous step."""
        self.keep_grads = keep_grads
        self.fused_call = fused_call      # one camo_forward_loss_backward instead of forward / loss / backward calls
        self.model = model
        self.engine = model._engine
        self.opt = FusedClipAdamW(model, lr=lr, weight_decay=weight_decay, max_norm=max_norm)
        self.grad_allreduce = grad_allreduce
        self.num_classes = model.config["num_classes"]
        self._grads_clean = False

    def step(self, rg_packed, nrs, kg, mask_label, edge_label, score_label, seed=None):
        """Returns (loss_terms [B,4], pred [B]) as device tensors; nothing is synchronised."""
        eng = self.engine
        batch = eng.make_batch(rg_packed, nrs, kg)
        ws = eng.workspace(batch)
        seed = eng.next_seed() if seed is None else seed
        training = self.model.training
        g = eng.ensure_flat_grads(attach=False)
        if self.keep_grads or not self._grads_clean:
            g.zero_()                                       # optimizer.zero_grad() per minibatch [:239]
        if self.fused_call:
            _, terms, pred = eng.train_raw(batch, ws, mask_label, edge_label, score_label, training, seed, eng._gtab)
        else:
            outs, _ = eng.forward_raw(batch, ws, training, seed)
            terms, d_pre, pred = multitask_loss(outs, mask_label, edge_label, score_label, self.num_classes, pre_activation=True)
            eng.backward_raw(batch, ws, outs, d_pre, training, seed, eng._gtab, pre_activation=True)
        self.opt.step(allreduce=self.grad_allreduce, zero_grads=not self.keep_grads)
        self._grads_clean = not self.keep_grads
        return terms, pred

    @torch.no_grad()
    def evaluate(self, rg_packed, nrs, kg):
        eng = self.engine
        batch = eng.make_batch(rg_packed, nrs, kg)
        outs, _ = eng.forward_raw(batch, eng.workspace(batch), False, 0)
        return outs


def train_epoch_fixed(model, dataloader, optimizer, device, epoch):
    """Reference signature [:223]; ``optimizer`` is a NativeTrainer (it owns the fused optimizer)."""
    trainer = optimizer
    model.train()
    terms_all, preds, labels = [], [], []
    for batch in dataloader:
        rg, nrs, kg, y, e, s = pack_samples(batch, device)
        terms, pred = trainer.step(rg, nrs, kg, y, e, s)
        terms_all.append(terms.sum(dim=1)); preds.append(pred); labels.append(y)
    losses = torch.cat(terms_all); preds = torch.cat(preds).cpu().long(); labels = torch.cat(labels).cpu()
    return float(losses.sum().item()) / max(len(preds), 1), calculate_f1_score(preds, labels)


def validate_fixed(model, dataloader, device):
    """Reference signature [:304]: (avg CE loss, f1 metrics, acc_0, acc_1)."""
    model.eval()
    eng = model._engine
    C = model.config["num_classes"]
    ce, preds, labels = [], [], []
    with torch.no_grad():
        for batch in dataloader:
            rg, nrs, kg, y, _, _ = pack_samples(batch, device)
            b = eng.make_batch(rg, nrs, kg)
            outs, _ = eng.forward_raw(b, eng.workspace(b), False, 0)
            logits = outs[:, :C]
            ce.append(torch.logsumexp(logits, dim=1) - logits.gather(1, y[:, None]).squeeze(1))
            preds.append(logits.argmax(dim=1)); labels.append(y)
    ce = torch.cat(ce); preds = torch.cat(preds).cpu(); labels = torch.cat(labels).cpu()
    f1 = calculate_f1_score(preds, labels)
    n0 = int((labels == 0).sum()); n1 = int((labels == 1).sum())
    acc_0 = 100 * int(((preds == labels) & (labels == 0)).sum()) / max(n0, 1)
    acc_1 = 100 * int(((preds == labels) & (labels == 1)).sum()) / max(n1, 1)
    return float(ce.sum().item()) / max(len(preds), 1), f1, acc_0, acc_1


def save_best_checkpoint(path, model, trainer, epoch, val_loss, val_f1, val_acc_0, val_acc_1, config):
    """The reference's checkpoint dict [:464-474]; loadable by its test script [test_multimodal.py:30-55]."""
    torch.save({
        "epoch": epoch,
        "model_state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
        "optimizer_state_dict": trainer.opt.state_dict(),
        "val_loss": val_loss,
        "val_f1_class_1": float(val_f1["f1_class_1"]),
        "val_f1_avg": float(val_f1["f1_avg"]),
        "val_acc_0": val_acc_0,
        "val_acc_1": val_acc_1,
        "config": config,
    }, path)


def fit(config, train_loader, val_loader, device="cuda", grad_allreduce=None, log=print):
    """Epoch loop of train_multimodal_fixed [:397-492] given ready data loaders (lists of
    sample dicts per minibatch).  Returns (model, history)."""
    model = build_multimodal_model(config["model"]).to(device)
    trainer = NativeTrainer(model, lr=config["learning_rate"], weight_decay=config["weight_decay"],
                            grad_allreduce=grad_allreduce)
    history = {k: [] for k in ("train_loss", "val_loss", "train_f1_class_0", "train_f1_class_1", "train_f1_avg",
                               "val_f1_class_0", "val_f1_class_1", "val_f1_avg", "val_acc_0", "val_acc_1")}
    best, patience, max_patience = 0.0, 0, 15
    os.makedirs(config["checkpoint_dir"], exist_ok=True)
    for epoch in range(config["epochs"]):
        trainer.opt.set_epoch(epoch)                         # CosineAnnealingWarmRestarts(T_0=10, T_mult=2) [:409-411]
        tl, tf1 = train_epoch_fixed(model, train_loader, trainer, device, epoch + 1)
        vl, vf1, a0, a1 = validate_fixed(model, val_loader, device)
        for k, v in (("train_loss", tl), ("val_loss", vl), ("train_f1_class_0", tf1["f1_class_0"]),
                     ("train_f1_class_1", tf1["f1_class_1"]), ("train_f1_avg", tf1["f1_avg"]),
                     ("val_f1_class_0", vf1["f1_class_0"]), ("val_f1_class_1", vf1["f1_class_1"]),
                     ("val_f1_avg", vf1["f1_avg"]), ("val_acc_0", a0), ("val_acc_1", a1)):
            history[k].append(float(v))
        log(f"Epoch {epoch + 1}/{config['epochs']} train loss {tl:.4f} F1_C1 {float(tf1['f1_class_1']):.3f} | "
            f"val loss {vl:.4f} F1_C1 {float(vf1['f1_class_1']):.3f} acc0 {a0:.1f}% acc1 {a1:.1f}%")
        if float(vf1["f1_class_1"]) > best:
            best, patience = float(vf1["f1_class_1"]), 0
            save_best_checkpoint(os.path.join(config["checkpoint_dir"], "multimodal_best_fixed.pth"), model, trainer,
                                 epoch, vl, vf1, a0, a1, config)
        else:
            patience += 1
            if patience >= max_patience:
                break
    with open(os.path.join(config["checkpoint_dir"], "training_history_fixed.json"), "w") as f:
        json.dump(history, f, indent=2)
    return model, history
