"""YAML -> flat attributes consumed as `mgr` (reference: configuration/config_manager.py:13-97).
Same five required sections (missing -> KeyError), same attribute names and defaults; the summary
print is kept behind `verbose`."""
from pathlib import Path

import yaml


class ConfigManager:
    def __init__(self, config_file, verbose=True):
        with open(config_file, "r") as f:
            config = yaml.safe_load(f)
        self.tr_info = config["tr_setup"]
        self.tr_configs = config["tr_config"]
        self.model_config = config["model_config"]
        self.dataset_config = config["dataset_config"]
        self.inference_config = config["inference_config"]
        self.verbose = verbose

        ti = self.tr_info
        self.model_name = ti.get("model_name", "Model")
        self.vram_max = float(ti.get("vram_max", 16))
        self.autoconfigure = bool(ti.get("autoconfigure", True))
        self.tr_val_split = float(ti.get("tr_val_split", 0.95))
        self.dilate_label = bool(ti.get("dilate_label", False))
        self.ckpt_out_base = Path(ti.get("ckpt_out_base", "./checkpoints/"))
        ckpt = ti.get("checkpoint_path", None)
        self.checkpoint_path = Path(ckpt) if ckpt else None
        self.load_weights_only = bool(ti.get("load_weights_only", False))
        self.tensorboard_log_dir = ti.get("tensorboard_log_dir", "./tensorboard_logs/")

        tc = self.tr_configs
        self.optimizer = tc.get("optimizer", "AdamW")
        self.initial_lr = float(tc.get("initial_lr", 1e-3))
        self.weight_decay = float(tc.get("weight_decay", 0))
        self.train_patch_size = tuple(tc.get("patch_size", [192, 192, 192]))
        self.train_batch_size = int(tc.get("batch_size", 2))
        self.gradient_accumulation = int(tc.get("gradient_accumulation", 1))
        self.max_steps_per_epoch = int(tc.get("max_steps_per_epoch", 500))
        self.max_val_steps_per_epoch = int(tc.get("max_val_steps_per_epoch", 25))
        self.train_num_dataloader_workers = int(tc.get("num_dataloader_workers", 4))
        self.max_epoch = int(tc.get("max_epoch", 500))

        dc = self.dataset_config
        self.min_labeled_ratio = float(dc.get("min_labeled_ratio", 0.1))
        self.min_bbox_percent = float(dc.get("min_bbox_percent", 0.95))
        self.use_cache = bool(dc.get("use_cache", True))
        self.cache_folder = Path(dc.get("cache_folder", "patch_cache"))
        self.in_channels = int(dc.get("in_channels", 1))
        self.tasks = dc.get("targets", {})
        self.volume_paths = dc.get("volume_paths", [])
        self.out_channels = tuple(info["channels"] for info in self.tasks.values())
        self.num_tasks = len(self.tasks)

        ic = self.inference_config
        self.infer_checkpoint_path = ic.get("checkpoint_path", None)
        self.infer_patch_size = tuple(ic.get("patch_size", self.train_patch_size))
        self.infer_batch_size = int(ic.get("batch_size", self.train_batch_size))
        self.infer_output_path = ic.get("output_path", "./outputs")
        if verbose:
            self._print_summary()

    def _print_summary(self):
        print("____________________________________________")
        for title, sec in (("Training Setup (tr_info)", self.tr_info), ("Training Config (tr_configs)", self.tr_configs),
                           ("Model Config (model_config)", self.model_config),
                           ("Dataset Config (dataset_config)", self.dataset_config),
                           ("Inference Config (inference_config)", self.inference_config)):
            print(f"{title}:")
            for k, v in sec.items():
                print(f"  {k}: {v}")
            print()
        print("____________________________________________")
