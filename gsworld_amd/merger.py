"""Scene assembly from GSWorld's ``configs/*.json``: load several Gaussian PLY files, attach
semantic labels, concatenate them into the one model the wrapper renders -- without ``plyfile`` and without
GSWorld's own Python.

Behavioural mirror of ``GaussianModelMerger`` (/root/reference/gsworld/utils/gaussian_merger.py): same class and
method names, same JSON schema (``{"models": [{"data_path", "semantic_labels", "transformation"}, ...]}``,
``:29-65``), same label rules (``:67-98``: a string is an ``.npy`` file of per-point labels, kept in its stored
dtype and given a trailing axis; a number labels every point; a tensor is used as it is; anything else means
zeros), same per-PLY loading through the semantic model's ``load_ply`` (``:162-191``; paths are relative to the
asset directory) and the same attribute-wise ``torch.cat`` (``:213-274``), so ``_opacity`` stays ``(N,1,1)`` and
``_semantics`` takes torch's promoted dtype when label sources differ.  The ``"transformation"`` entry is read and
ignored, as in the reference (its ``apply_transformation`` is never called).

Pinned by ``tests/golden/merger.npz``: the merged tensors the reference class itself produced for a tiny PLY pair
(``tools/make_golden.py``).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

from . import ply

_ATTRS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


def _gaussian_model_class():
    """``scene.gaussian_model.GaussianModel`` as GSWorld resolves it (``sys.path.append(GS_DIR)``); falls back to this
    package's own ``gs_compat`` directory when no 3DGS python layer is on the path yet."""
    try:
        from scene.gaussian_model import GaussianModel
    except ImportError:
        compat = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gs_compat")
        if compat not in sys.path:
            sys.path.append(compat)
        from scene.gaussian_model import GaussianModel
    return GaussianModel


_SEMANTIC_CLASS = None


def semantic_model_class():
    """The counterpart of ``Semantic3DGSWrapper`` (semantic_3dgs_wrapper.py:34-184) the merger instantiates: the 3DGS
    ``GaussianModel`` plus a ``_semantics (N,1)`` tensor, loaded / saved through :mod:`gsworld_amd.ply`."""
    global _SEMANTIC_CLASS
    if _SEMANTIC_CLASS is not None:
        return _SEMANTIC_CLASS
    Base = _gaussian_model_class()

    class SemanticGaussianModel(Base):
        def __init__(self, sh_degree, optimizer_type="default"):
            super().__init__(sh_degree, optimizer_type)
            self._semantics = torch.empty(0)

        @property
        def get_semantics(self):
            return self._semantics

        def construct_list_of_attributes(self):
            return super().construct_list_of_attributes() + ["semantics"]

        def load_ply(self, path, use_train_test_exp=False, device="cuda"):
            # frozen tensors, (N,1,1) opacity, semantics column or zeros: semantic_3dgs_wrapper.py:100-167
            ply.read_gaussian_ply(path, self, device=device)
            self.active_sh_degree = self.max_sh_degree

        def save_ply(self, path):
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            ply.write_gaussian_ply(path, self, with_semantics=True)

    _SEMANTIC_CLASS = SemanticGaussianModel
    return SemanticGaussianModel


class GaussianModelMerger:
    def __init__(self, device="cuda", asset_dir: str | None = None, sh_degree: int = 3):
        """``asset_dir``: the directory ``data_path`` / label paths are relative to (the reference's ``ASSET_DIR``,
        /root/reference/gsworld/constants.py:8); default: ``$GSWORLD_ASSET_DIR`` or the current directory."""
        self.device = device if (device != "cuda" or torch.cuda.is_available()) else "cpu"
        self.asset_dir = asset_dir if asset_dir is not None else os.environ.get("GSWORLD_ASSET_DIR", os.getcwd())
        self.sh_degree = sh_degree
        self.models = []
        self.model_paths = []
        self.model_configs = []
        self.merged_model = None

    # ---- configuration ---------------------------------------------------------------------------------------
    def load_config_from_json(self, json_path):
        if not os.path.exists(json_path):
            raise FileNotFoundError(f"JSON configuration file not found: {json_path}")
        try:
            with open(json_path, "r") as f:
                config = json.load(f)
        except json.JSONDecodeError:
            raise ValueError(f"Invalid JSON format in file: {json_path}") from None
        entries = config.get("models") if isinstance(config, dict) else None
        if not isinstance(entries, list):
            raise ValueError("JSON file should contain a 'models' list")
        self.model_configs = list(entries)
        return self.model_configs

    def assign_semantic_labels(self, model_data, semantic_labels):
        n = model_data._xyz.shape[0]
        if isinstance(semantic_labels, str):
            if not os.path.exists(semantic_labels):
                raise AssertionError(f"semantic label file not found: {semantic_labels}")
            model_data._semantics = torch.from_numpy(np.load(semantic_labels)).to(self.device)[..., None]
        elif isinstance(semantic_labels, (int, float)):
            model_data._semantics = torch.full((n, 1), float(semantic_labels), device=self.device)
        elif isinstance(semantic_labels, torch.Tensor):
            model_data._semantics = semantic_labels.to(self.device)
        else:
            model_data._semantics = torch.zeros(n, 1, device=self.device)
        return model_data

    # ---- loading ---------------------------------------------------------------------------------------------
    def load_model_from_config(self, model_config):
        if "data_path" not in model_config:
            raise ValueError("Missing required 'data_path' in model config")
        ply_path = os.path.join(self.asset_dir, model_config["data_path"])
        if not os.path.exists(ply_path):
            raise FileNotFoundError(f"PLY file not found: {ply_path}")
        labels = model_config.get("semantic_labels", None)
        model = semantic_model_class()(self.sh_degree)
        model.load_ply(ply_path, device=self.device)
        if labels is not None:
            if isinstance(labels, str):
                labels = os.path.join(self.asset_dir, labels)
            model = self.assign_semantic_labels(model, labels)
        self.models.append(model)
        self.model_paths.append(ply_path)
        return len(self.models) - 1

    def load_multiple_models(self, model_configs):
        return [self.load_model_from_config(cfg) for cfg in model_configs]

    def load_models_from_config(self, json_path):
        return self.load_multiple_models(self.load_config_from_json(json_path))

    def get_model(self, index):
        if 0 <= index < len(self.models):
            return self.models[index]
        raise IndexError(f"Model index {index} is out of range")

    # ---- merging ---------------------------------------------------------------------------------------------
    def merge_models(self, indices=None):
        if not self.models:
            raise ValueError("No models to merge")
        chosen = self.models if indices is None else [self.get_model(i) for i in indices]
        if not chosen:
            raise ValueError("No valid models to merge")
        merged = semantic_model_class()(self.sh_degree)
        for attr in _ATTRS:
            setattr(merged, attr, torch.cat([getattr(m, attr) for m in chosen], dim=0))
        merged._semantics = torch.cat(
            [m._semantics if hasattr(m, "_semantics") else torch.zeros(m._xyz.shape[0], 1, device=self.device)
             for m in chosen], dim=0)
        merged.active_sh_degree = merged.max_sh_degree
        self.merged_model = merged
        return merged

    def save_merged_model(self, output_path):
        if self.merged_model is None:
            raise ValueError("No merged model exists. Call merge_models() first.")
        self.merged_model.save_ply(output_path)
        return True

    def get_merged_model(self):
        if self.merged_model is None:
            raise ValueError("No merged model exists. Call merge_models() first.")
        return self.merged_model

    def clear_models(self):
        self.models, self.model_paths, self.model_configs = [], [], []


def merge_scene(config_json: str, asset_dir: str | None = None, device="cuda"):
    """``configs/<scene>.json`` -> the merged semantic model (what ``gaussian_merger.main(path)`` returns)."""
    merger = GaussianModelMerger(device=device, asset_dir=asset_dir)
    merger.load_models_from_config(config_json)
    merged = merger.merge_models()
    merger.clear_models()
    return merged
