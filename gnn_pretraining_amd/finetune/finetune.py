"""Fine-tuning loop on libgnnmp with the structure of src/finetune/finetune.py: FinetuneConfig, process_batch per task
type, one optimiser step per batch (AdamW over FinetuneGNN.param_groups), best-validation checkpoint.  Data are
synthetic stand-ins of the reference's shapes (Cora-like single graph for *_NC, ENZYMES-shaped batches for graph
classification); sklearn metrics and wandb are out of scope.  Link-prediction fine-tuning needs the
hard-negative miner and is not part of this round."""
from __future__ import annotations

import argparse
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import torch

from .. import operators as O, synthetic as S
from ..constants import DOMAIN_DIMENSIONS, NUM_CLASSES, TASK_TYPES
from ..graph import Batch
from ..models.finetune_model import FinetuneGNN, create_finetune_model

OUTPUT_DIR = Path(__file__).resolve().parents[2] / "outputs" / "finetune"
BATCH_SIZES = {"ENZYMES": 32, "PTC_MR": 32, "Cora_NC": -1, "CiteSeer_NC": -1, "Cora_LP": 256, "CiteSeer_LP": 256}
EPOCHS = {"ENZYMES": 100, "PTC_MR": 100, "Cora_NC": 200, "CiteSeer_NC": 200, "Cora_LP": 300, "CiteSeer_LP": 300}
PATIENCE_FRACTION = 0.5


@dataclass
class FinetuneConfig:
    domain_name: str
    finetune_strategy: str
    pretrained_scheme: str
    seed: int
    exp_name: str = None
    task_type: str = None
    batch_size: int = None
    epochs: int = None
    patience: int = None

    def __post_init__(self) -> None:
        self.exp_name = f"{self.domain_name}_{self.finetune_strategy}_{self.pretrained_scheme}"
        self.task_type = TASK_TYPES[self.domain_name]
        self.batch_size = BATCH_SIZES[self.domain_name]
        self.epochs = EPOCHS[self.domain_name]
        self.patience = int(self.epochs * PATIENCE_FRACTION)


def classification_loss(logits: torch.Tensor, targets: torch.Tensor, num_classes: int) -> torch.Tensor:
    """finetune.py:150-158 / 169-177: CE (mean), or BCE-with-logits on logits[:,1] for binary domains."""
    if num_classes == 2:
        return torch.nn.functional.binary_cross_entropy_with_logits(logits[:, 1], targets.float())
    return O.cross_entropy_sum(logits, targets) / targets.numel()


def process_batch(model: FinetuneGNN, batch, device, task_type: str, domain_name: str):
    if task_type == "graph_classification":
        b = batch.to(device)
        logits = model(b)
        return classification_loss(logits, b.y, NUM_CLASSES[domain_name]), logits
    if task_type == "node_classification":
        data, idx, targets = batch
        logits = O.take_rows(model(data), idx)
        return classification_loss(logits, targets, NUM_CLASSES[domain_name]), logits
    raise NotImplementedError("link-prediction fine-tuning (hard-negative miner) is a later row of SURVEY section 8f")


def finetune(cfg: FinetuneConfig, epochs: Optional[int] = None, device: Optional[str] = None) -> float:
    torch.manual_seed(cfg.seed)
    gen = torch.Generator().manual_seed(cfg.seed)
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("the HIP path needs a GPU (no CPU fallback)")
        device = "cuda"
    dev = torch.device(device)
    model = create_finetune_model(dev, cfg)
    opt = torch.optim.AdamW(model.param_groups)
    dim, ncls = DOMAIN_DIMENSIONS[cfg.domain_name], NUM_CLASSES[cfg.domain_name]
    if cfg.task_type == "node_classification":
        n = 2708 if cfg.domain_name.startswith("Cora") else 3327
        g = S.cora_like(gen, num_nodes=n, undirected_edges=5429 if n == 2708 else 4552, dim=dim, num_classes=ncls)
        data = Batch.from_data_list([g]).to(dev)
        perm = torch.randperm(n, generator=gen)
        y = g.y.to(dev)
        train = [(data, perm[:20 * ncls].to(dev), y[perm[:20 * ncls]])]
        val = [(data, perm[20 * ncls:20 * ncls + 500].to(dev), y[perm[20 * ncls:20 * ncls + 500]])]
    else:
        mk = lambda k: [Batch.from_data_list([S.random_graph(gen, dim, num_classes=ncls) for _ in range(cfg.batch_size)]) for _ in range(k)]
        train, val = mk(15), mk(2)
    OUTPUT_DIR.mkdir(parents=True, exist_ok=True)
    path = OUTPUT_DIR / f"model_{cfg.exp_name}_{cfg.seed}.pt"
    best, stale, t0 = -1.0, 0, time.time()
    for epoch in range(1, (epochs or cfg.epochs) + 1):
        model.train()
        for batch in train:
            loss, _ = process_batch(model, batch, dev, cfg.task_type, cfg.domain_name)
            opt.zero_grad()
            loss.backward()
            opt.step()
        model.eval()
        correct = total = 0
        with torch.no_grad():
            for batch in val:
                _, logits = process_batch(model, batch, dev, cfg.task_type, cfg.domain_name)
                tgt = batch[2] if cfg.task_type == "node_classification" else batch.y.to(dev)
                correct += int((logits.argmax(1) == tgt).sum())
                total += tgt.numel()
        acc = correct / max(total, 1)
        if acc > best:
            best, stale = acc, 0
            torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "val_metrics": {"val/accuracy": acc}}, path)
        else:
            stale += 1
        if stale >= cfg.patience:
            break
    print(f"{cfg.exp_name}: best val accuracy {best:.4f}, training_time {time.time() - t0:.2f}s, saved {path}")
    return best


def main() -> None:
    p = argparse.ArgumentParser()
    p.add_argument("--domain_name", required=True)
    p.add_argument("--finetune_strategy", required=True, choices=["full_finetune", "linear_probe"])
    p.add_argument("--pretrained_scheme", required=True)
    p.add_argument("--seed", type=int, required=True)
    p.add_argument("--epochs", type=int, default=None)
    p.add_argument("--device", type=str, default=None)
    a = p.parse_args()
    finetune(FinetuneConfig(a.domain_name, a.finetune_strategy, a.pretrained_scheme, a.seed), a.epochs, a.device)


if __name__ == "__main__":
    main()
