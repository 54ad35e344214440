"""Drop-in for the reference's ``test.py`` missing-ratio sweep on the MI355X hot path (reference test.py:15-171).

Same flags (:15-40), same flow: load ``<model_ckpt_dir>/<datasetName>_<fusion_type>.pth`` (the file ``train_ddp.train`` writes),
for every ``test_type`` - ``concat_median`` / ``concat_mean`` first run the encoder over the training set and hand the
per-modality median / mean embedding to ``fusion.set_statistics`` (:97-115) - then evaluate every missing-modality scenario at
every missing ratio and write ``new_txt_experiment/<dataset>_<test_type>_<scenario>.txt`` in the reference's format (:118-169).
The dataset loaders are injected (``src/dataset/data_loader.py`` reads private datasets): ``train_loader`` as in ``train_ddp``,
``test_loader[scenario][ratio]`` -> iterable of ``(data, label, missing_index)``.
"""
from __future__ import annotations

import argparse
import os
from typing import Dict, Iterable, Optional

import numpy as np
import torch
from torch import nn

from .languagebind import LanguageBind
from .nn import HipCrossEntropyLoss
from .src.model.baseline import finetune_model
from .train_ddp import _metrics, _prepare, set_seed

_csv = lambda s: s.split(",")   # noqa: E731  (the reference declares these as type=list; a comma list is the usable form)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--train_mode", type=str, default="classification")
    p.add_argument("--datasetName", type=str, default="eNTERFACE")
    p.add_argument("--csv_path", type=str, default="")
    p.add_argument("--modality_types", type=_csv, default=["video", "audio"])
    p.add_argument("--test_missing_type", type=_csv, default=["video", "audio", "mixed"])
    p.add_argument("--model_ckpt_dir", type=str, default="./final_model")
    p.add_argument("--feature_dims", type=int, default=768)
    p.add_argument("--fusion_type", type=str, default="sum")
    p.add_argument("--test_types", type=_csv, default=["sum"])
    p.add_argument("--fusion_dim", type=int, default=256)
    p.add_argument("--dropout_prob", type=float, default=0.1)
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--device", type=str, default="cuda")
    return p.parse_args(argv)


def get_criterion():
    return HipCrossEntropyLoss()


def calculate_statistics(data_list, type):
    """reference test.py:54-61"""
    data_array = np.concatenate(data_list, axis=0)
    return (np.mean(data_array, axis=0) if type == "mean" else np.median(data_array, axis=0)).tolist()


def training_set_statistics(model: nn.Module, train_loader: Iterable, modality_types, kind: str, device) -> Dict[str, list]:
    """reference test.py:98-114: encoder embeddings of the whole training set -> per-modality mean / median row"""
    emb = {m: [] for m in modality_types}
    with torch.no_grad():
        for data, _, _ in train_loader:
            out = model.encoder(_prepare(data, device))
            for m in modality_types:
                emb[m].append(out[m].detach().cpu().numpy())
    return {m: calculate_statistics(emb[m], kind) for m in modality_types}


def test(args, train_loader: Iterable, test_loader: Dict[str, Dict[object, Iterable]], output_dims: int,
         encoder_model: Optional[nn.Module] = None, compute_dtype: torch.dtype = torch.bfloat16, log=print):
    set_seed(args.seed)
    out_dir = "./new_txt_experiment"
    os.makedirs(out_dir, exist_ok=True)
    if encoder_model is None:
        clip_type = {m: f"LanguageBind_{m.capitalize()}" for m in args.modality_types if m != "language"}
        encoder_model = LanguageBind(clip_type=clip_type, cache_dir="./cache_dir", compute_dtype=compute_dtype)
    model = finetune_model(args, output_dims, encoder_model)
    ckpt = torch.load(os.path.join(args.model_ckpt_dir, f"{args.datasetName}_{args.fusion_type}.pth"), map_location="cpu", weights_only=False)
    model.load_state_dict(ckpt["model_state_dict"])
    model = model.to(args.device)
    model.eval()
    criterion = get_criterion()
    metrics = None
    for test_type in args.test_types:
        if test_type in ("concat_median", "concat_mean"):
            stats = training_set_statistics(model, train_loader, args.modality_types, "median" if test_type == "concat_median" else "mean",
                                            args.device)
            model.fusion.set_statistics(stats, args.modality_types)
        for scenario in args.test_missing_type:
            name = f"{args.datasetName}_{test_type}_{scenario}"
            with open(f"{out_dir}/{name}.txt", "w", encoding="utf-8") as fout:
                for ratio, loader in test_loader[scenario].items():
                    log(f"Testing with missing ratio: {ratio}")
                    total, nb, probs, preds, labels_all = 0.0, 0, [], [], []
                    with torch.no_grad():
                        for data, label, missing_index in loader:
                            labels = (label["label"] if isinstance(label, dict) else label).to(args.device)
                            outputs = model(_prepare(data, args.device), missing_index.to(args.device))
                            outputs = outputs[1] if isinstance(outputs, tuple) else outputs
                            total += float(criterion(outputs, labels))
                            nb += 1
                            probs.append(torch.softmax(outputs.float(), dim=-1).cpu().numpy())
                            preds.append(outputs.argmax(dim=1).cpu().numpy())
                            labels_all.append(labels.cpu().numpy())
                    metrics = _metrics(np.concatenate(labels_all), np.concatenate(preds), np.concatenate(probs))
                    metrics["loss"] = total / max(nb, 1)
                    fout.write(f"Testing with missing ratio: {ratio}\nTest Results:\nTest Loss: {metrics['loss']:.4f}\n"
                               f"Test Accuracy: {metrics['accuracy']:.4f}\nTest F1 Score: {metrics['f1']:.4f}\n"
                               f"Test AUC: {metrics['auc']:.4f}\n\n")
    return model, metrics
