"""Thin DDP training / validation loop that replaces the Lightning ``Trainer`` on this path
(reference: ``scripts/train.py:30`` -> ``CustomLightningCLI`` -> ``pl.Trainer.fit`` with
``DDPStrategy(find_unused_parameters=False)``, cli.py:48).

One process per GPU (launch with ``python -m torch.distributed.run --nproc-per-node N ...``);
``batch_size`` is per process exactly as under Lightning DDP, so scaling is weak by construction.
Per optimizer step there is ONE collective: a sum all-reduce (RCCL over xGMI; backend "nccl" is
RCCL on ROCm) of the flat fp32 gradient buffer (5.36 MB for the LFO-net), after which the fused
AdamW kernel applies ``grad / world_size``.  Metrics follow Lightning's ``on_epoch=True,
sync_dist=True``: epoch means, averaged across ranks.
"""
import os
import time
from typing import Any, Callable, Dict, List, Optional

import torch
import torch.distributed as dist

from .optim import FlatAdamW


def dist_env() -> Dict[str, int]:
    return {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
            "world_size": int(os.environ.get("WORLD_SIZE", "1"))}


def init_distributed(backend: Optional[str] = None) -> Dict[str, int]:
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    env = dist_env()
    if env["world_size"] > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(env["local_rank"])
        dist.init_process_group(backend=backend, rank=env["rank"], world_size=env["world_size"])
    return env


def allreduce_flat_grad(flat_grad: torch.Tensor, world_size: int) -> float:
    """Sum all-reduce of the flat gradient; returns the scale the optimizer must apply (1/world).
    Every rank must call this every step -- a rank with no valid clip contributes zeros."""
    if world_size > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        return 1.0 / world_size
    return 1.0


def reduce_metrics(logged: Dict[str, List[torch.Tensor]], world_size: int) -> Dict[str, float]:
    """Epoch means of the logged scalars, averaged over ranks (Lightning sync_dist=True)."""
    names = sorted(logged.keys())
    if not names:
        return {}
    vals = torch.stack([torch.stack([v.float().reshape(()) for v in logged[n]]).mean() for n in names])
    if world_size > 1:
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
        vals = vals / world_size
    return {n: float(v) for n, v in zip(names, vals.cpu())}


class Trainer:
    def __init__(self, max_epochs: int = 1, limit_train_batches: Optional[int] = None,
                 limit_val_batches: Optional[int] = None, num_sanity_val_steps: int = 0,
                 log_fn: Optional[Callable[[str], None]] = print, checkpoints: Optional["CheckpointKeeper"] = None,
                 **ignored: Any) -> None:
        self.max_epochs = max_epochs
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.num_sanity_val_steps = num_sanity_val_steps
        self.log_fn = log_fn
        self.checkpoints = checkpoints
        self.env = dist_env()
        self.history: List[Dict[str, float]] = []

    def _say(self, msg: str) -> None:
        if self.log_fn is not None and self.env["rank"] == 0:
            self.log_fn(msg)

    def train_step(self, module, optimizer: FlatAdamW, batch) -> Optional[torch.Tensor]:
        """forward + loss + backward + all-reduce + AdamW for an LFOExtraction-style module."""
        optimizer.zero_grad()
        loss = module.training_step(batch, 0)
        if loss is not None:
            loss.backward()
        scale = allreduce_flat_grad(optimizer.flat_grad, self.env["world_size"])
        optimizer.step(grad_scale=scale)
        return loss

    def validate(self, module, datamodule, n_steps: Optional[int] = None) -> Dict[str, float]:
        module.eval()
        module.logged.clear()
        n = n_steps or datamodule.val_steps_per_epoch()
        if self.limit_val_batches is not None:
            n = min(n, self.limit_val_batches)
        for i in range(n):
            module.validation_step(datamodule.val_batch(), i)
        out = reduce_metrics(module.logged, self.env["world_size"])
        module.logged.clear()
        return out

    def fit(self, module, datamodule, optimizer: FlatAdamW) -> List[Dict[str, float]]:
        manual = getattr(module, "automatic_optimization", True) is False
        if self.num_sanity_val_steps:
            self.validate(module, datamodule, self.num_sanity_val_steps)
        for epoch in range(self.max_epochs):
            module.train()
            module.logged.clear()
            n = datamodule.train_steps_per_epoch()
            if self.limit_train_batches is not None:
                n = min(n, self.limit_train_batches)
            t0 = time.time()
            for i in range(n):
                batch = datamodule.train_batch()
                if manual:
                    module.training_step(batch, i, optimizer=optimizer, world_size=self.env["world_size"])
                else:
                    self.train_step(module, optimizer, batch)
            metrics = reduce_metrics(module.logged, self.env["world_size"])
            metrics.update(self.validate(module, datamodule))
            metrics["epoch"] = epoch
            metrics["epoch_time_s"] = time.time() - t0
            self.history.append(metrics)
            if self.checkpoints is not None:
                step = getattr(optimizer, "step_count", 0)
                self.checkpoints.update(module, optimizer, epoch, step, metrics, rank=self.env["rank"])
            self._say(" ".join(f"{k}={v:.5g}" for k, v in metrics.items()))
        return self.history


# ---- checkpoints (reference: cli.py:29-37,145-150 ModelCheckpoint(monitor="val/loss", save_top_k=1,
# save_last=True), filename "{model_name}__{dataset_name}__epoch_{e}_step_{s}"; lightning.py:237-241 and
# scripts/extract_model_weights.py:38-47 for the consumers of the format) --------------------------------
def save_checkpoint(path: str, module, optimizer: Optional[FlatAdamW], epoch: int, global_step: int,
                    extra: Optional[Dict[str, Any]] = None) -> None:
    """Lightning-compatible layout: ``state_dict`` keyed like the LightningModule (``model.`` /
    ``effect_model.`` / ``lfo_model.`` prefixes come from the attribute names), ``epoch``,
    ``global_step``; the flat AdamW state goes under ``optimizer_states``."""
    blob = {"state_dict": {k: v.detach().cpu() for k, v in module.state_dict().items()}, "epoch": epoch,
            "global_step": global_step, "pytorch-lightning_version": "mod_extraction_amd"}
    if optimizer is not None:
        blob["optimizer_states"] = [{k: (v.cpu() if isinstance(v, torch.Tensor) else v)
                                     for k, v in optimizer.state_dict().items()}]
    if extra:
        blob.update(extra)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(blob, path)


def extract_model_weights(ckpt_path: str, out_path: str, prefix: str = "model.") -> Dict[str, torch.Tensor]:
    """scripts/extract_model_weights.py:38-47: bare state dict of one sub-module (prefix stripped)."""
    sd = torch.load(ckpt_path, map_location="cpu")["state_dict"]
    bare = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    torch.save(bare, out_path)
    return bare


class CheckpointKeeper:
    """save_top_k=1 on ``val/loss`` (mode min) + save_last, with the reference's file-name rule."""

    def __init__(self, dirpath: str, model_name: str = "local_model", dataset_name: str = "local_dataset") -> None:
        self.dirpath, self.model_name, self.dataset_name = dirpath, model_name, dataset_name
        self.best, self.best_path = float("inf"), None

    def name(self, epoch: int, step: int) -> str:
        return f"{self.model_name}__{self.dataset_name}__epoch_{epoch}_step_{step}.ckpt"

    def update(self, module, optimizer, epoch: int, step: int, metrics: Dict[str, float], rank: int = 0) -> None:
        if rank != 0:
            return
        save_checkpoint(os.path.join(self.dirpath, "last.ckpt"), module, optimizer, epoch, step)
        v = metrics.get("val/loss")
        if v is not None and v < self.best:
            if self.best_path and os.path.exists(self.best_path):
                os.remove(self.best_path)
            self.best, self.best_path = v, os.path.join(self.dirpath, self.name(epoch, step))
            save_checkpoint(self.best_path, module, optimizer, epoch, step)
