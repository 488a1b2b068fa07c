"""Hooks of the training loop (ponder/engines/hooks/misc.py): IterationTimer :33-76,
InformationWriter :79-144, CheckpointSaver :147-205, CheckpointLoader :208-253.  The evaluator
hooks of the default runtime are registered as inert entries: pre-training sets evaluate=False."""
import json
import os
import time
from collections import OrderedDict

import torch

from ..utils import comm
from ..utils.registry import Registry

HOOKS = Registry("hooks")


class HookBase:
    trainer = None

    def before_train(self): pass
    def before_epoch(self): pass
    def before_step(self): pass
    def after_step(self): pass
    def after_epoch(self): pass
    def after_train(self): pass


@HOOKS.register_module()
class IterationTimer(HookBase):
    def __init__(self, warmup_iter=1):
        self._warmup_iter = warmup_iter
        self._t = time.perf_counter()
        self._seen = 0
        self.batch_times, self.data_times = [], []

    def before_epoch(self):
        self._t = time.perf_counter()

    def before_step(self):
        self.trainer.comm_info["data_time"] = time.perf_counter() - self._t

    def after_step(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()  # step time includes the device work of this step
        now = time.perf_counter()
        batch_time = now - self._t
        self._t = now
        self._seen += 1
        info = self.trainer.comm_info
        info["batch_time"] = batch_time
        if self._seen > self._warmup_iter:
            self.batch_times.append(batch_time)
            self.data_times.append(info["data_time"])
        avg = sum(self.batch_times) / max(len(self.batch_times), 1)
        remain = (self.trainer.max_iter - info["global_iter"] - 1) * avg
        info["iter_info"] = info.get("iter_info", "") + (
            f"Data {info['data_time']:.3f} Batch {batch_time:.3f} "
            f"Remain {int(remain // 3600):02d}:{int(remain % 3600 // 60):02d}:{int(remain % 60):02d} ")


@HOOKS.register_module()
class InformationWriter(HookBase):
    """Logs every scalar of the model's output dict.  The scalars are fetched with ONE stacked
    device->host copy per step (the reference calls .item() per key: 9 syncs per step)."""

    def __init__(self, log_every=1):
        self.log_every = log_every
        self.history = []

    def before_step(self):
        info = self.trainer.comm_info
        info["iter_info"] = (f"Train: [{self.trainer.epoch + 1}/{self.trainer.max_epoch}]"
                             f"[{info['iter'] + 1}/{len(self.trainer.train_loader)}] ")

    def after_step(self):
        info = self.trainer.comm_info
        out = info.get("model_output_dict", {})
        keys = [k for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1]
        vals = torch.stack([out[k].detach().float().reshape(()) for k in keys]).tolist() if keys else []
        scalars = dict(zip(keys, vals))
        lr = self.trainer.optimizer.param_groups[0]["lr"]
        self.history.append(dict(iter=info["global_iter"], lr=lr, **scalars))
        if info["global_iter"] % self.log_every == 0:
            msg = info.get("iter_info", "") + " ".join(f"{k}: {v:.4f}" for k, v in scalars.items())
            self.trainer.logger.info(msg + f" Lr: {lr:.5f}")
        w = self.trainer.writer
        if w is not None:
            w.write(json.dumps(self.history[-1]) + "\n")
            w.flush()


@HOOKS.register_module()
class CheckpointSaver(HookBase):
    def __init__(self, save_freq=None):
        self.save_freq = save_freq

    def after_epoch(self):
        if not comm.is_main_process():
            return
        t = self.trainer
        path = os.path.join(t.cfg.save_path, "model", "model_last.pth")
        model = t.model.module if hasattr(t.model, "module") else t.model
        state = dict(epoch=t.epoch + 1, state_dict=model.state_dict(),
                     optimizer=t.optimizer.state_dict(), scheduler=t.scheduler.state_dict(),
                     scaler=t.scaler.state_dict() if t.scaler is not None else None,
                     best_metric_value=t.best_metric_value)
        torch.save(state, path + ".tmp")
        os.replace(path + ".tmp", path)  # atomic
        if self.save_freq and (t.epoch + 1) % self.save_freq == 0:
            torch.save(state, os.path.join(t.cfg.save_path, "model", f"epoch_{t.epoch + 1}.pth"))
        t.logger.info(f"Saved checkpoint to {path}")


@HOOKS.register_module()
class CheckpointLoader(HookBase):
    def __init__(self, keywords="", replacement=None, strict=False):
        self.keywords = keywords
        self.replacement = replacement if replacement is not None else keywords
        self.strict = strict

    def before_train(self):
        t = self.trainer
        weight = t.cfg.get("weight")
        if not weight:
            t.logger.info("No weight found, training from scratch")
            return
        if not os.path.isfile(weight):
            t.logger.info(f"No weight found at: {weight}")
            return
        ckpt = torch.load(weight, map_location="cpu", weights_only=False)
        wrapped = hasattr(t.model, "module")
        weights = OrderedDict()
        for k, v in ckpt["state_dict"].items():
            k = k[7:] if k.startswith("module.") else k       # normalise the DDP prefix
            if self.keywords and self.keywords in k:
                k = k.replace(self.keywords, self.replacement)
            weights[("module." + k) if wrapped else k] = v
        info = t.model.load_state_dict(weights, strict=self.strict)
        t.logger.info(f"Loaded weight from {weight}; missing keys: {list(info.missing_keys)[:8]}")
        if t.cfg.get("resume"):
            t.start_epoch = ckpt["epoch"]
            t.best_metric_value = ckpt.get("best_metric_value", t.best_metric_value)
            t.optimizer.load_state_dict(ckpt["optimizer"])
            t.scheduler.load_state_dict(ckpt["scheduler"])
            if t.scaler is not None and ckpt.get("scaler") is not None:
                t.scaler.load_state_dict(ckpt["scaler"])


class _Inert(HookBase):
    def __init__(self, **kwargs):
        pass


for _name in ("SemSegEvaluator", "PreciseEvaluator", "ClsEvaluator", "InsSegEvaluator",
              "DataCacheOperator", "RuntimeProfiler", "RuntimeProfilerV2"):
    HOOKS.register_module(name=_name, module=type(_name, (_Inert,), {}))
