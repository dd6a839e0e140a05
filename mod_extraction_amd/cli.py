"""Drop-in for the reference's entry layer (mod_extraction/cli.py:21-167 ``CustomLightningCLI`` on top
of jsonargparse / LightningCLI, neither of which is needed here): parses the same YAML schema and
drives ``trainer.Trainer`` instead of ``pl.Trainer``.

Supported schema (what the shipped configs use):
  * ``class_path`` / ``init_args`` objects, nested arbitrarily; ``mod_extraction.*`` class paths resolve to
    this package's mirrors (``mod_extraction_amd.*``), ``torch.optim.AdamW`` to the flat HIP AdamW
  * a value may be a path to another YAML (``model: ../configs/models/spectral_2dcnn.yml``); paths are
    tried relative to the current directory (the reference runs from ``scripts/``) and to the config file
  * the "link if possible" rules of configs/cli_config.yml:21-45 (n_samples / sr from data to the models)
  * ``seed_everything``, ``trainer.{max_epochs, num_sanity_val_steps, limit_*_batches}``, ``custom.*``,
    ``ckpt_path`` (Lightning ``.ckpt`` or bare ``.pt`` state dict, prefixes stripped like
    scripts/extract_model_weights.py:38-47)
GPU-only: there is no CPU fallback, so the reference's CPU overrides (cli.py:128-143) do not apply.
"""
import copy
import importlib
import logging
import os
from typing import Any, Dict, List, Optional

import torch
import yaml

log = logging.getLogger(__name__)

CLASS_ALIASES = {"torch.optim.AdamW": "mod_extraction_amd.optim.FlatAdamW"}
LINKS = [  # (source path, destination path) -- configs/cli_config.yml:21-45
    ("data.init_args.n_samples", "model.init_args.model.init_args.n_samples"),
    ("data.init_args.n_samples", "model.init_args.lfo_model.init_args.n_samples"),
    ("data.init_args.n_samples", "model.init_args.param_model.init_args.n_samples"),
    ("data.init_args.shared_args.n_samples", "model.init_args.model.init_args.n_samples"),
    ("data.init_args.shared_args.n_samples", "model.init_args.lfo_model.init_args.n_samples"),
    ("data.init_args.shared_args.n_samples", "model.init_args.param_model.init_args.n_samples"),
    ("data.init_args.sr", "model.init_args.sr"),
    ("data.init_args.shared_args.sr", "model.init_args.sr"),
    ("data.init_args.sr", "model.init_args.model.init_args.sr"),
    ("data.init_args.shared_args.sr", "model.init_args.model.init_args.sr"),
    ("data.init_args.sr", "model.init_args.lfo_model.init_args.sr"),
    ("data.init_args.shared_args.sr", "model.init_args.lfo_model.init_args.sr"),
]


def _is_yaml_path(v: Any) -> bool:
    return isinstance(v, str) and v.lower().endswith((".yml", ".yaml"))


def _find(path: str, base_dirs: List[str]) -> Optional[str]:
    for b in base_dirs:
        p = os.path.normpath(os.path.join(b, path))
        if os.path.isfile(p):
            return p
    return None


def load_config(path: str) -> Dict[str, Any]:
    """Load a YAML config and inline every value that is itself a path to a YAML file."""
    path = os.path.abspath(path)
    base_dirs = [os.getcwd(), os.path.dirname(path), os.path.join(os.path.dirname(path), "..", "scripts")]

    def expand(node: Any) -> Any:
        if _is_yaml_path(node):
            found = _find(node, base_dirs)
            if found is None:
                raise FileNotFoundError(f"config indirection {node!r} not found relative to {base_dirs}")
            with open(found) as f:
                return expand(yaml.safe_load(f))
        if isinstance(node, dict):
            return {k: expand(v) for k, v in node.items()}
        if isinstance(node, list):
            return [expand(v) for v in node]
        return node

    with open(path) as f:
        return expand(yaml.safe_load(f))


def _get(cfg: Dict[str, Any], dotted: str) -> Any:
    cur: Any = cfg
    for k in dotted.split("."):
        if not isinstance(cur, dict) or k not in cur:
            return None
        cur = cur[k]
    return cur


def _set_if_possible(cfg: Dict[str, Any], dotted: str, value: Any) -> bool:
    """cli.py:71-103 (link_arguments_if_possible): the destination receives the source value when it is reachable and
    NOT already given -- a value written in the YAML stays (configs/models/baseline_rand_lfo.yml keeps its 345 frames at
    172.5 Hz although the data module runs 88200 samples at 44.1 kHz; the reference logs "overriding" there but its
    assignment sits in the other branch)."""
    keys = dotted.split(".")
    cur: Any = cfg
    for k in keys[:-1]:
        if not isinstance(cur, dict) or k not in cur or not isinstance(cur[k], dict):
            return False
        cur = cur[k]
    if keys[-1] in cur and cur[keys[-1]] != value:
        log.info("link %s: destination already set to %r, keeping it (source %r)", dotted, cur[keys[-1]], value)
        return False
    cur[keys[-1]] = value
    return True


def apply_links(cfg: Dict[str, Any]) -> Dict[str, Any]:
    for src, dst in LINKS:
        v = _get(cfg, src)
        if v is not None:
            _set_if_possible(cfg, dst, v)
    return cfg


def resolve_class(class_path: str):
    class_path = CLASS_ALIASES.get(class_path, class_path)
    if class_path.startswith("mod_extraction."):
        class_path = "mod_extraction_amd." + class_path[len("mod_extraction."):]
    mod, _, name = class_path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(spec: Any, **extra: Any) -> Any:
    """Build the object graph of a ``class_path`` / ``init_args`` spec (depth first)."""
    if isinstance(spec, dict) and "class_path" in spec:
        kwargs = {k: instantiate(v) for k, v in (spec.get("init_args") or {}).items()}
        kwargs.update(extra)
        return resolve_class(spec["class_path"])(**kwargs)
    if isinstance(spec, dict):
        return {k: instantiate(v) for k, v in spec.items()}
    if isinstance(spec, list):
        return [instantiate(v) for v in spec]
    return spec


def load_weights(module: torch.nn.Module, path: str) -> None:
    """Lightning ``.ckpt`` (``state_dict`` with ``model.`` / ``effect_model.`` / ``lfo_model.`` prefixes) or a
    bare ``.pt`` state dict."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    sd = blob.get("state_dict", blob) if isinstance(blob, dict) else blob
    own = module.state_dict()
    if not set(sd).issubset(set(own)):          # bare sub-module weights: find the attribute they belong to
        for prefix in ("model.", "effect_model.", "lfo_model."):
            cand = {prefix + k: v for k, v in sd.items()}
            if set(cand).issubset(set(own)):
                sd = cand
                break
    missing, unexpected = module.load_state_dict(sd, strict=False)
    if unexpected:
        raise KeyError(f"unexpected keys in {path}: {unexpected[:5]}")
    # a bare sub-module file legitimately leaves the other sub-modules alone (e.g. LSTM weights into a module
    # that also holds the frozen LFO net); anything missing INSIDE a sub-module the file does cover is an error
    covered = {k.split(".")[0] for k in sd}
    bad = [k for k in missing if k.split(".")[0] in covered]
    if bad:
        raise KeyError(f"checkpoint {path} lacks {len(bad)} keys of the sub-modules it covers: {bad[:5]}")
    if missing:
        log.info("checkpoint %s leaves %d keys of other sub-modules untouched", path, len(missing))


def _next_version_dir(root: str) -> str:
    """TensorBoardLogger(save_dir="lightning_logs", name=None) of the reference (cli.py:39-45):
    ``lightning_logs/version_<n>``, checkpoints under ``checkpoints/``."""
    n = 0
    if os.path.isdir(root):
        taken = [int(d[8:]) for d in os.listdir(root) if d.startswith("version_") and d[8:].isdigit()]
        n = max(taken) + 1 if taken else 0
    return os.path.join(root, f"version_{n}")


def seed_everything(seed: int) -> None:
    import random
    import numpy as np
    random.seed(seed); np.random.seed(seed % (2 ** 32)); torch.manual_seed(seed)


class CustomLightningCLI:
    """``CustomLightningCLI(args=["fit" | "validate", "-c", config.yml])`` as in scripts/train.py:30 and
    scripts/validate.py:25.  With ``run=False`` only the object graph is built (``.model``,
    ``.datamodule``, ``.optimizer_spec``, ``.trainer``).

    ``fit`` installs the reference's ModelCheckpoint policy (cli.py:29-37,145-150: best ``val/loss`` + last,
    ``{model_name}__{dataset_name}__epoch_{e}_step_{s}.ckpt`` under ``lightning_logs/version_N/checkpoints``);
    ``ckpt_path`` must exist (Lightning raises too) unless ``allow_missing_ckpt`` -- the pretrained blobs of the
    reference are not part of its repository -- and, for ``fit``, is a resume (optimizer state, epoch, step).
    Seeding: the model is built under the common ``seed_everything`` value so that all DDP replicas start identical;
    the host RNGs that drive the data stream are then re-seeded with ``seed + rank`` (Lightning gives each rank its
    own stream through the distributed sampler and per-worker seeds)."""

    # cli.py:22-50: the reference keeps accelerator / callbacks / logger here; this loop has its checkpoint policy and
    # logger built in, so scripts that pass `trainer_defaults=CustomLightningCLI.trainer_defaults` keep working
    trainer_defaults: Dict[str, Any] = {}

    def __init__(self, args: List[str], trainer_defaults: Optional[Dict[str, Any]] = None, run: bool = True,
                 device: Optional[torch.device] = None, allow_missing_ckpt: Optional[bool] = None,
                 log_dir: str = "lightning_logs") -> None:
        assert args
        if args[0] in ("fit", "validate"):
            self.subcommand = args[0]
        else:                                    # LightningCLI(run=False): build the object graph only, no subcommand
            assert not run, "a subcommand (fit | validate) is required unless run=False"
            self.subcommand = None
        cfg_path = args[args.index("-c") + 1] if "-c" in args else args[args.index("--config") + 1]
        self.config = apply_links(load_config(cfg_path))
        if "--ckpt_path" in args:                # command-line override, as `validate --config c --ckpt_path p`
            self.config["ckpt_path"] = args[args.index("--ckpt_path") + 1]
        self.seed = self.config.get("seed_everything")
        if self.seed is not None:
            seed_everything(int(self.seed))
        from . import trainer as tr
        self.env = tr.init_distributed()
        if device is None:
            device = torch.device("cuda", self.env["local_rank"])
        self.device = device
        self.custom = self.config.get("custom", {}) or {}
        keys = ("max_epochs", "num_sanity_val_steps", "limit_train_batches", "limit_val_batches")
        tkw = dict(trainer_defaults or {})
        tkw.update({k: v for k, v in (self.config.get("trainer") or {}).items() if k in keys})
        tkw = {k: v for k, v in tkw.items() if k in keys + ("log_fn", "checkpoints")}
        if self.subcommand == "fit" and run and "checkpoints" not in tkw:
            tkw["checkpoints"] = tr.CheckpointKeeper(
                os.path.join(_next_version_dir(log_dir), "checkpoints"),
                self.custom.get("model_name", "local_model"), self.custom.get("dataset_name", "local_dataset"),
                hyper_parameters={"config": copy.deepcopy(self.config)})
        self.trainer = tr.Trainer(**tkw)
        self.datamodule = instantiate(self.config["data"])
        self.model = instantiate(self.config["model"]).to(device)
        self.optimizer_spec = self.config.get("optimizer")
        self.optimizer = None
        self.ckpt_file = None
        ckpt = self.config.get("ckpt_path")
        if allow_missing_ckpt is None:
            allow_missing_ckpt = os.environ.get("MODEX_ALLOW_MISSING_CKPT", "0") == "1"
        if ckpt:
            found = _find(ckpt, [os.getcwd(), os.path.dirname(os.path.abspath(cfg_path))])
            if found:
                load_weights(self.model, found)
                self.ckpt_file = found
            elif allow_missing_ckpt:
                log.warning("ckpt_path %s not found (large blobs are not part of the repository): continuing with "
                            "freshly initialised weights because allow_missing_ckpt is set", ckpt)
            else:
                raise FileNotFoundError(f"ckpt_path {ckpt!r} not found (pass allow_missing_ckpt=True or set "
                                        f"MODEX_ALLOW_MISSING_CKPT=1 to run with random weights)")
        if run:
            self.run()

    def prepare_data_stream(self) -> None:
        """Per-rank host RNG stream (parameter draws, chunk search, SpecAugment masks) + data-module setup."""
        seed = int(self.seed) if self.seed is not None else 43
        rank = self.env["rank"]
        seed_everything(seed + rank)
        self.datamodule.setup(self.device, rank=rank, seed=seed)

    def run(self):
        from . import trainer as tr
        rank = self.env["rank"]
        self.prepare_data_stream()
        if self.subcommand == "validate":
            self.model.eval()
            metrics = self.trainer.validate(self.model, self.datamodule)
            if rank == 0:
                print(metrics)
            return metrics
        params = [p for p in self.model.parameters() if p.requires_grad]
        self.optimizer = instantiate(self.optimizer_spec, params=params)
        if self.env["world_size"] > 1:          # replicas must start identical whatever the RNG history
            torch.distributed.broadcast(self.optimizer.flat_param, src=0)
        if self.ckpt_file is not None:
            tr.resume_from_checkpoint(self.ckpt_file, self.model, self.optimizer, self.trainer)
        return self.trainer.fit(self.model, self.datamodule, self.optimizer)
