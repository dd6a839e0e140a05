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
    """Rank layout from the torchrun environment.  ``MODEX_SHARE_GPU=1`` (testing the multi-rank code paths on a box
    with fewer GPUs than ranks; use it with ``MODEX_DIST_BACKEND=gloo`` -- RCCL refuses two ranks on one device) maps
    LOCAL_RANK onto the devices that exist."""
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MODEX_SHARE_GPU") == "1" and torch.cuda.device_count() > 0:
        local %= torch.cuda.device_count()
    return {"rank": int(os.environ.get("RANK", "0")), "local_rank": local,
            "world_size": int(os.environ.get("WORLD_SIZE", "1"))}


def init_distributed(backend: Optional[str] = None) -> Dict[str, int]:
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    env = dist_env()
    if env["world_size"] > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("MODEX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (the only mode this pool's driver supports)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(env["local_rank"])
        dist.init_process_group(backend=backend, rank=env["rank"], world_size=env["world_size"])
    return env


class CollectiveTimer:
    """Optional device timing of the gradient all-reduce (bench.py's N > 1 line: `allreduce_ms`).  Inside
    ``with CollectiveTimer() as ct:`` every ``allreduce_flat_grad`` call is bracketed by a HIP event pair on the
    stream the collective is enqueued on; ``ct.results_ms()`` synchronises and returns the durations."""

    active: Optional["CollectiveTimer"] = None

    def __init__(self) -> None:
        self.pairs: List[Any] = []

    def __enter__(self) -> "CollectiveTimer":
        CollectiveTimer.active = self
        return self

    def __exit__(self, *exc) -> None:
        CollectiveTimer.active = None

    def results_ms(self) -> List[float]:
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in self.pairs]


def allreduce_flat_grad(flat_grad: torch.Tensor, world_size: int) -> float:
    """Sum all-reduce of the flat gradient; returns the scale the optimizer must apply (1/world).
    Every rank must call this every step -- a rank with no valid clip contributes zeros."""
    if world_size > 1:
        ct = CollectiveTimer.active
        if ct is not None and flat_grad.is_cuda:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
            b.record()
            ct.pairs.append((a, b))
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        return 1.0 / world_size
    return 1.0


def reduce_metrics(logged: Dict[str, List[torch.Tensor]], world_size: int,
                   names: Optional[List[str]] = None) -> Dict[str, float]:
    """Epoch means of the logged scalars, averaged over the ranks that logged them (Lightning
    ``on_epoch=True, sync_dist=True``).  Under DDP ``names`` must be the same list on every rank (the
    trainer derives it from the module's ``loss_dict``): a rank that logged nothing for a name -- a TBPTT
    rank whose every batch had no valid LFO -- still takes part in the collective with count 0, so the
    all-reduce shapes always match."""
    if names is None:
        if world_size > 1:
            raise ValueError("reduce_metrics under DDP needs the fixed metric-name list")
        names = sorted(logged.keys())
    if not names:
        return {}
    dev = next((v[0].device for v in logged.values() if v), None)
    if dev is None or (world_size > 1 and dist.get_backend() == "nccl" and dev.type != "cuda"):
        # a rank that logged nothing still needs a tensor the backend can reduce (RCCL: on its GPU)
        dev = torch.device("cuda", torch.cuda.current_device()) if world_size > 1 and dist.get_backend() == "nccl" \
            else torch.device("cpu")
    sums = torch.zeros(2, len(names), dtype=torch.float64, device=dev)
    for i, n in enumerate(names):
        vals = [v.double().reshape(()) for v in logged.get(n, []) if v is not None]
        if vals:
            sums[0, i] = torch.stack(vals).mean()
            sums[1, i] = 1.0
    if world_size > 1:
        if dist.get_backend() == "gloo":
            sums = sums.cpu()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    sums = sums.cpu()
    return {n: float(sums[0, i] / sums[1, i]) for i, n in enumerate(names) if float(sums[1, i]) > 0}


def metric_names(module, prefix: str) -> List[str]:
    """The scalars a BaseLightingModule logs per step under ``prefix`` (lightning.py:33-62)."""
    return [f"{prefix}/{k}" for k in getattr(module, "loss_dict", {})] + [f"{prefix}/loss"]


def _limit(n: int, limit) -> int:
    """Lightning's ``limit_*_batches``: an int caps the count, a float in [0, 1] is a fraction of it."""
    if limit is None:
        return n
    if isinstance(limit, float) and not float(limit).is_integer():
        assert 0.0 <= limit <= 1.0, "fractional limit_*_batches must be in [0, 1]"
        return int(n * limit)
    if isinstance(limit, float) and limit == 1.0:
        return n
    return min(n, int(limit))


class Trainer:
    def __init__(self, max_epochs: int = 1, limit_train_batches=None, limit_val_batches=None, num_sanity_val_steps: int = 0,
                 log_fn: Optional[Callable[[str], None]] = print, checkpoints: Optional["CheckpointKeeper"] = None,
                 **ignored: Any) -> None:
        self.max_epochs = max_epochs
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.num_sanity_val_steps = num_sanity_val_steps
        self.log_fn = log_fn
        self.checkpoints = checkpoints
        self.env = dist_env()
        self.history: List[Dict[str, float]] = []
        self.start_epoch = 0            # > 0 after resume_from_checkpoint

    def _say(self, msg: str) -> None:
        if self.log_fn is not None and self.env["rank"] == 0:
            self.log_fn(msg)

    def train_step(self, module, optimizer: FlatAdamW, batch) -> Optional[torch.Tensor]:
        """forward + loss + backward + all-reduce + AdamW for an LFOExtraction-style module."""
        optimizer.zero_grad()
        loss = module.training_step(batch, 0)
        if loss is not None:
            with optimizer.direct_backward():       # first backward after zero_grad(): gradients may be written in place
                loss.backward()
        elif self.env["world_size"] == 1:
            return None                 # Lightning skips the optimizer step when training_step returns None
        scale = allreduce_flat_grad(optimizer.flat_grad, self.env["world_size"])   # DDP: stay in lock-step
        optimizer.step(grad_scale=scale)
        return loss

    def validate(self, module, datamodule, n_steps: Optional[int] = None) -> Dict[str, float]:
        module.eval()
        module.logged.clear()
        n = n_steps or _limit(datamodule.val_steps_per_epoch(), self.limit_val_batches)
        # validation_step recomputes prepare() itself: a prefetch hook left installed by fit() would run the frozen
        # extractor a second time per batch on the side stream and throw the result away
        pause = getattr(datamodule, "pause_ahead", None)
        if pause is not None:
            pause(True)
        try:
            for i in range(n):
                module.validation_step(datamodule.val_batch(), i)
        finally:
            if pause is not None:
                pause(False)
        out = reduce_metrics(module.logged, self.env["world_size"], metric_names(module, "val"))
        module.logged.clear()
        return out

    def fit(self, module, datamodule, optimizer: FlatAdamW) -> List[Dict[str, float]]:
        manual = getattr(module, "automatic_optimization", True) is False
        if self.num_sanity_val_steps:
            self.validate(module, datamodule, self.num_sanity_val_steps)
        # (an extractor that is being trained must see the weights of the step that uses it: no look-ahead then)
        prefetch = (manual and hasattr(module, "prepare_ahead") and hasattr(datamodule, "set_ahead_fn")
                    and getattr(module, "freeze_lfo_model", True))
        main = None
        if prefetch:
            datamodule.set_ahead_fn(module.prepare_ahead)        # frozen-extractor forward one batch ahead (side stream)
            # the latency-bound recurrence and the prefetch work on disjoint CUs (streams.py)
            from . import streams
            dev = next(module.parameters()).device
            part = streams.cu_partition(dev, main_workgroups=getattr(datamodule, "batch_size", 0)) if dev.type == "cuda" and hasattr(datamodule, "use_side_stream") else None
            if part is not None:
                main = part[0]
                datamodule.use_side_stream(part[1])
                main.wait_stream(torch.cuda.current_stream(dev))
        if main is None:
            self._fit_epochs(module, datamodule, optimizer, manual, prefetch)
        else:
            with torch.cuda.stream(main):
                self._fit_epochs(module, datamodule, optimizer, manual, prefetch)
            torch.cuda.current_stream(main.device).wait_stream(main)
        return self.history

    def _fit_epochs(self, module, datamodule, optimizer: FlatAdamW, manual: bool, prefetch: bool) -> None:
        for epoch in range(self.start_epoch, self.max_epochs):
            module.train()
            module.logged.clear()
            n = _limit(datamodule.train_steps_per_epoch(), self.limit_train_batches)
            t0 = time.time()
            for i in range(n):
                batch = datamodule.train_batch()
                if manual:
                    kw = {"prep": datamodule.take_ahead()} if prefetch else {}
                    module.training_step(batch, i, optimizer=optimizer, world_size=self.env["world_size"], **kw)
                else:
                    self.train_step(module, optimizer, batch)
            metrics = reduce_metrics(module.logged, self.env["world_size"], metric_names(module, "train"))
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
LIGHTNING_VERSION = "2.0.2"         # requirements_pipchill.txt: pytorch-lightning==2.0.2; Lightning's
#                                     migrate_checkpoint parses this field with packaging.Version


def adamw_state_dict(optimizer: FlatAdamW) -> Dict[str, Any]:
    """The flat optimizer state in ``torch.optim.AdamW.state_dict()`` layout (one entry per parameter, in
    ``optimizer.params`` order = the order the reference hands ``model.parameters()`` to AdamW), so that the
    reference -- or plain torch -- can ``load_state_dict`` it."""
    state, off = {}, 0
    for i, p in enumerate(optimizer.params):
        k = p.numel()
        state[i] = {"step": torch.tensor(float(optimizer.step_count)),
                    "exp_avg": optimizer.exp_avg[off:off + k].view(p.shape).detach().cpu().clone(),
                    "exp_avg_sq": optimizer.exp_avg_sq[off:off + k].view(p.shape).detach().cpu().clone()}
        off += k
    group = {"lr": optimizer.lr, "betas": tuple(optimizer.betas), "eps": optimizer.eps,
             "weight_decay": optimizer.weight_decay, "amsgrad": False, "maximize": False, "foreach": None,
             "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(optimizer.params)))}
    return {"state": state, "param_groups": [group]}


def load_adamw_state_dict(optimizer: FlatAdamW, sd: Dict[str, Any]) -> None:
    """Inverse of ``adamw_state_dict`` (also accepts a state dict written by ``torch.optim.AdamW`` itself)."""
    state = sd["state"]
    assert len(state) in (0, len(optimizer.params)), "optimizer state does not match the parameter list"
    off, step = 0, 0
    for i, p in enumerate(optimizer.params):
        k = p.numel()
        if i in state:
            st = state[i]
            optimizer.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
            optimizer.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            step = int(float(st["step"]))
        off += k
    optimizer.step_count = step
    g = sd["param_groups"][0]
    optimizer.lr, optimizer.betas = float(g["lr"]), (float(g["betas"][0]), float(g["betas"][1]))
    optimizer.eps, optimizer.weight_decay = float(g["eps"]), float(g["weight_decay"])


def save_checkpoint(path: str, module, optimizer: Optional[FlatAdamW], epoch: int, global_step: int,
                    extra: Optional[Dict[str, Any]] = None, callbacks: Optional[Dict[str, Any]] = None,
                    hyper_parameters: Optional[Dict[str, Any]] = None) -> None:
    """A checkpoint the reference's Lightning 2.0.2 stack can read back (``Trainer.validate/fit(ckpt_path=)``,
    ``scripts/extract_model_weights.py``): ``state_dict`` keyed like the LightningModule (``model.`` /
    ``effect_model.`` / ``lfo_model.`` prefixes come from the attribute names), ``epoch``, ``global_step``,
    a real ``pytorch-lightning_version``, ``optimizer_states`` in torch AdamW layout, ``lr_schedulers``,
    ``callbacks`` (the ModelCheckpoint bookkeeping) and ``hyper_parameters``."""
    blob = {"epoch": epoch, "global_step": global_step, "pytorch-lightning_version": LIGHTNING_VERSION,
            "state_dict": {k: v.detach().cpu() for k, v in module.state_dict().items()},
            "callbacks": callbacks or {}, "optimizer_states": [], "lr_schedulers": [],
            "hparams_name": "kwargs", "hyper_parameters": hyper_parameters or {}}
    # no "loops" entry: Lightning restores loop progress only when the key is present, and falls back to
    # epoch / global_step otherwise
    if optimizer is not None:
        blob["optimizer_states"] = [adamw_state_dict(optimizer)]
    if extra:
        blob.update(extra)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(blob, path)


def resume_from_checkpoint(path: str, module, optimizer: Optional[FlatAdamW], trainer: Optional["Trainer"] = None
                           ) -> Dict[str, Any]:
    """``fit`` with ``ckpt_path`` is a resume under Lightning: weights, optimizer moments and step count,
    epoch.  Bare ``.pt`` state dicts carry weights only."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    if not (isinstance(blob, dict) and "state_dict" in blob):
        return {}
    if optimizer is not None and blob.get("optimizer_states"):
        load_adamw_state_dict(optimizer, blob["optimizer_states"][0])
    if trainer is not None:
        trainer.start_epoch = int(blob.get("epoch", -1)) + 1
        if trainer.checkpoints is not None:
            trainer.checkpoints.load_state(blob.get("callbacks") or {})
    return blob


def extract_model_weights(ckpt_path: str, out_path: str, prefix: str = "model.") -> Dict[str, torch.Tensor]:
    """scripts/extract_model_weights.py:38-47: bare state dict of one sub-module (prefix stripped)."""
    sd = torch.load(ckpt_path, map_location="cpu")["state_dict"]
    bare = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    torch.save(bare, out_path)
    return bare


class CheckpointKeeper:
    """save_top_k=1 on ``val/loss`` (mode min) + save_last, with the reference's file-name rule."""

    STATE_KEY = ("ModelCheckpoint{'monitor': 'val/loss', 'mode': 'min', 'every_n_train_steps': 0, "
                 "'every_n_epochs': 1, 'train_time_interval': None}")          # Lightning's callback state_key

    def __init__(self, dirpath: str, model_name: str = "local_model", dataset_name: str = "local_dataset",
                 hyper_parameters: Optional[Dict[str, Any]] = None) -> None:
        self.dirpath, self.model_name, self.dataset_name = dirpath, model_name, dataset_name
        self.best, self.best_path = float("inf"), None
        self.hyper_parameters = hyper_parameters

    def name(self, epoch: int, step: int) -> str:
        return f"{self.model_name}__{self.dataset_name}__epoch_{epoch}_step_{step}.ckpt"

    def state(self, current: Optional[float] = None) -> Dict[str, Any]:
        last = os.path.join(self.dirpath, "last.ckpt")
        # the best SCORE is written whenever one exists, also when its file belongs to an earlier run (best_path None after
        # a resume into a new directory): a resume of that resume must still know what it has to beat
        have = self.best != float("inf")
        return {self.STATE_KEY: {"monitor": "val/loss", "best_model_score": torch.tensor(self.best) if have else None,
                                 "best_model_path": self.best_path or "", "current_score": None if current is None else torch.tensor(current),
                                 "dirpath": self.dirpath, "best_k_models": {} if self.best_path is None else {self.best_path: torch.tensor(self.best)},
                                 "kth_best_model_path": self.best_path or "", "kth_value": torch.tensor(self.best),
                                 "last_model_path": last}}

    def _owns(self, path: Optional[str]) -> bool:
        """True for a file inside THIS run's checkpoint directory (the only files this keeper may delete)."""
        if not path:
            return False
        mine = os.path.realpath(self.dirpath)
        return os.path.commonpath([mine, os.path.realpath(path)]) == mine

    def load_state(self, callbacks: Dict[str, Any]) -> None:
        """Resume: the best SCORE always carries over (from `best_model_score`, else from a finite `kth_value`); the best
        PATH only when the checkpoint was written into this keeper's own directory.  The CLI resumes into a fresh
        `version_N+1/checkpoints`; the earlier run's files (often the very `ckpt_path` being resumed) are never this
        run's to delete.  Lightning 2.0.2's ModelCheckpoint differs both ways when `dirpath` changed -- it keeps
        `best_model_path` and drops `best_model_score` / `best_k_models`, so its resumed run saves its first epoch as
        "best" whatever the score; keeping the score is the deliberate deviation here."""
        for key, st in callbacks.items():
            if not str(key).startswith("ModelCheckpoint"):
                continue
            score = st.get("best_model_score")
            if score is None and st.get("kth_value") is not None and float(st["kth_value"]) != float("inf"):
                score = st["kth_value"]
            if score is not None:
                self.best = float(score)
                path = st.get("best_model_path") or None
                same_dir = os.path.realpath(str(st.get("dirpath") or "")) == os.path.realpath(self.dirpath)
                self.best_path = path if (same_dir and self._owns(path)) else None

    def update(self, module, optimizer, epoch: int, step: int, metrics: Dict[str, float], rank: int = 0) -> None:
        if rank != 0:
            return
        v = metrics.get("val/loss")
        improved = v is not None and v < self.best
        if improved:
            if self._owns(self.best_path) and os.path.exists(self.best_path):
                os.remove(self.best_path)
            self.best, self.best_path = v, os.path.join(self.dirpath, self.name(epoch, step))
        kw = dict(callbacks=self.state(v), hyper_parameters=self.hyper_parameters)
        if improved:
            save_checkpoint(self.best_path, module, optimizer, epoch, step, **kw)
        save_checkpoint(os.path.join(self.dirpath, "last.ckpt"), module, optimizer, epoch, step, **kw)
