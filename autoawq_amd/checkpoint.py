"""AWQ checkpoints on disk <-> WQLinear_* modules (the data format
either side of the hot path).

What an AutoAWQ checkpoint is (reference, under /root/reference):
  * `config.json` carries `quantization_config = {"quant_method": "awq", "bits", "group_size",
    "zero_point", "version", "modules_to_not_convert"}` -- `awq/models/_config.py:84-102`
    (`to_transformers_dict` / `from_transformers_dict`), read back at `_config.py:58-71`;
  * the weights are the model's `state_dict()` written as (sharded) safetensors --
    `awq/models/base.py:274-319`; the quantised Linears contribute the BUFFERS
    `<name>.qweight / .qzeros / .scales (/ .bias)` in the layout of their `version`;
  * loading = build the fp16 skeleton, replace every nn.Linear of the decoder layers (minus
    `modules_to_not_convert`) by `WQLinear_*.from_linear(..., init_only=True)`, then load the
    state dict -- `base.py:640-681` (`_load_quantized_modules`) and `base.py:527-535`.

This module does exactly that for any `nn.Module` skeleton, with no dependency on the reference's
model zoo, and adds the inverse (`save_quantized`) plus a round-to-nearest packer
(`quantize_linears_rtn`, the arithmetic of `awq/quantize/quantizer.py:74-109,228-262` without the
AWQ scale search, which is out of scope) so that checkpoints can be produced for tests.
"""
import json
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .modules.linear import WQLinear_GEMM, WQLinear_GEMV, WQLinear_GEMVFast

VERSIONS = {"gemm": WQLinear_GEMM, "gemv": WQLinear_GEMV, "gemv_fast": WQLinear_GEMVFast}
QUANT_BUFFERS = ("qweight", "qzeros", "scales")


@dataclass
class AwqConfig:
    """Same fields, defaults and dict forms as `awq/models/_config.py:9-102`."""
    quant_method: str = "awq"
    zero_point: bool = True
    q_group_size: int = 128
    w_bit: int = 4
    version: str = "gemm"
    modules_to_not_convert: Optional[List[str]] = None
    config_file_name = "config.json"

    @classmethod
    def from_dict(cls, quant_config: Optional[Dict] = None):
        if not quant_config:
            return cls()
        cfg = cls(**quant_config)
        cfg.version = cfg.version.lower()
        return cfg

    @staticmethod
    def from_transformers_dict(d: Dict) -> Dict:
        return {"quant_method": d.get("quant_method"), "zero_point": d.get("zero_point"),
                "q_group_size": d.get("group_size"), "w_bit": d.get("bits"), "version": d.get("version"),
                "modules_to_not_convert": d.get("modules_to_not_convert")}

    @classmethod
    def from_pretrained(cls, save_dir: str):
        """Local directories only (no hub access in this build): reads `quantization_config` of
        `config.json`; a missing file or key gives the defaults, like the reference."""
        path = os.path.join(save_dir, cls.config_file_name)
        if os.path.exists(path):
            with open(path, "r", encoding="utf-8") as f:
                loaded = json.load(f)
            qc = loaded.get("quantization_config")
            if qc is not None:
                cfg = cls(**cls.from_transformers_dict(qc))
                cfg.version = cfg.version.lower()
                return cfg
        return cls()

    def to_dict(self) -> Dict:
        return {"zero_point": self.zero_point, "q_group_size": self.q_group_size, "w_bit": self.w_bit,
                "version": self.version, "modules_to_not_convert": self.modules_to_not_convert}

    def to_transformers_dict(self) -> Dict:
        return {"quant_method": self.quant_method, "zero_point": self.zero_point, "group_size": self.q_group_size,
                "bits": self.w_bit, "version": self.version.lower(),
                "modules_to_not_convert": self.modules_to_not_convert}


# ---- module surgery (awq/utils/module.py:11-65)
def get_named_linears(module: nn.Module) -> Dict[str, nn.Linear]:
    return {name: m for name, m in module.named_modules() if isinstance(m, nn.Linear)}


def set_op_by_name(layer: nn.Module, name: str, new_module: nn.Module) -> None:
    levels = name.split(".")
    mod = layer
    for lvl in levels[:-1]:
        mod = mod[int(lvl)] if lvl.isdigit() else getattr(mod, lvl)
    setattr(mod, levels[-1], new_module)


def exclude_layers_to_not_quantize(linears: Dict[str, nn.Linear], modules_to_not_convert) -> Dict[str, nn.Linear]:
    if modules_to_not_convert is None:
        return linears
    return {n: m for n, m in linears.items() if not any(key in n for key in modules_to_not_convert)}


def find_decoder_layers(model: nn.Module) -> nn.ModuleList:
    """The reference asks each model class (`get_model_layers`, e.g. `awq/models/llama.py`); the
    common attribute paths are tried here, then the largest ModuleList that contains Linears."""
    for path in ("model.layers", "model.decoder.layers", "transformer.h", "transformer.blocks", "gpt_neox.layers",
                 "model.language_model.layers", "language_model.model.layers", "layers"):
        obj = model
        try:
            for a in path.split("."):
                obj = getattr(obj, a)
        except AttributeError:
            continue
        if isinstance(obj, nn.ModuleList):
            return obj
    best = None
    for m in model.modules():
        if isinstance(m, nn.ModuleList) and any(isinstance(x, nn.Linear) for x in m.modules()):
            if best is None or len(m) > len(best):
                best = m
    if best is None:
        raise ValueError("no decoder layer list found: pass layers= explicitly")
    return best


def _linear_class(version: str):
    try:
        return VERSIONS[version.lower()]
    except KeyError:
        raise ValueError(f"unsupported AWQ version {version!r}: this build serves {sorted(VERSIONS)} "
                         "(Marlin / ExLlama / IPEX repacks are other kernels' formats)") from None


def replace_quantized_linears(model: nn.Module, quant_config: AwqConfig, layers=None) -> List[str]:
    """`_load_quantized_modules` (`awq/models/base.py:640-681`): every nn.Linear inside the decoder
    layers, except `modules_to_not_convert`, becomes an empty WQLinear_* of the checkpoint's version.
    Returns the replaced module names (relative to `model`)."""
    if quant_config.w_bit != 4:
        raise NotImplementedError("Only 4-bit are supported for now.")
    cls = _linear_class(quant_config.version)
    layers = find_decoder_layers(model) if layers is None else layers
    prefix = {id(m): n for n, m in model.named_modules()}
    replaced = []
    for layer in layers:
        named = exclude_layers_to_not_quantize(get_named_linears(layer), quant_config.modules_to_not_convert)
        for name, lin in named.items():
            q = cls.from_linear(lin, quant_config.w_bit, quant_config.q_group_size, True)
            set_op_by_name(layer, name, q)
            base = prefix.get(id(layer), "")
            replaced.append(f"{base}.{name}" if base else name)
    return replaced


# ---- reading / writing the tensor files
def _checkpoint_files(path: str) -> List[str]:
    if os.path.isfile(path):
        return [path]
    index = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(index):
        with open(index, "r", encoding="utf-8") as f:
            weight_map = json.load(f)["weight_map"]
        return [os.path.join(path, fn) for fn in sorted(set(weight_map.values()))]
    single = os.path.join(path, "model.safetensors")
    if os.path.exists(single):
        return [single]
    files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no .safetensors file in {path}")
    return [os.path.join(path, f) for f in files]


def read_state_dict(path: str, device="cpu") -> Dict[str, torch.Tensor]:
    from safetensors import safe_open

    state = {}
    for fn in _checkpoint_files(path):
        with safe_open(fn, framework="pt", device=str(device)) as f:
            for k in f.keys():
                if k in state:
                    raise ValueError(f"tensor {k} appears in more than one shard")
                state[k] = f.get_tensor(k)
    return state


def load_quantized(model: nn.Module, path: str, quant_config: Optional[AwqConfig] = None, device=None, layers=None,
                   strict: bool = True, repack: Optional[str] = None):
    """Skeleton `model` (fp16 nn.Linear everywhere) + checkpoint directory -> quantised model.
    Shapes and dtypes of every quantised buffer are checked against the module built from the
    skeleton (a wrong `group_size` / `version` fails here, not inside a kernel).  `repack="gemm"`
    (or "gemv" / "gemv_fast") converts the loaded modules to another layout after loading."""
    if quant_config is None:
        quant_config = AwqConfig.from_pretrained(path if os.path.isdir(path) else os.path.dirname(path))
    replaced = replace_quantized_linears(model, quant_config, layers)
    state = read_state_dict(path)
    want = model.state_dict()
    for name in replaced:
        for buf in QUANT_BUFFERS:
            key = f"{name}.{buf}"
            if key not in state:
                raise KeyError(f"checkpoint has no tensor {key} (version {quant_config.version!r}, "
                               f"group_size {quant_config.q_group_size})")
            if state[key].shape != want[key].shape or state[key].dtype != want[key].dtype:
                raise ValueError(f"{key}: checkpoint {tuple(state[key].shape)} {state[key].dtype} vs module "
                                 f"{tuple(want[key].shape)} {want[key].dtype} built for version "
                                 f"{quant_config.version!r}, group_size {quant_config.q_group_size}")
    result = model.load_state_dict(state, strict=False, assign=True)
    tied = set(getattr(model, "_tied_weights_keys", None) or [])
    missing = [k for k in result.missing_keys if k not in tied and not k.endswith("rotary_emb.inv_freq")]
    if strict and (missing or result.unexpected_keys):
        raise KeyError(f"load_quantized: missing {missing[:8]} unexpected {list(result.unexpected_keys)[:8]}")
    if hasattr(model, "tie_weights"):
        model.tie_weights()
    if device is not None:
        model.to(device)
    if repack is not None and repack.lower() != quant_config.version:
        # declared, optional: serve a checkpoint of one format with the kernels of another
        # (integer repack on the device, bit-exact: autoawq_amd/utils/convert.py)
        from .utils.convert import convert_model

        convert_model(model, repack)
    return model, quant_config


def save_quantized(model: nn.Module, quant_config: AwqConfig, save_dir: str, shard_size="5GB") -> None:
    """`save_quantized` (`awq/models/base.py:274-319`): config.json with `quantization_config` in
    the transformers form + the state dict as (sharded) safetensors."""
    os.makedirs(save_dir, exist_ok=True)
    cfg = getattr(model, "config", None)
    if cfg is not None and hasattr(cfg, "to_dict"):
        d = cfg.to_dict()
    else:
        d = {}
    d["quantization_config"] = quant_config.to_transformers_dict()
    with open(os.path.join(save_dir, AwqConfig.config_file_name), "w", encoding="utf-8") as f:
        json.dump(d, f, indent=2, sort_keys=True, default=str)
    state = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    from huggingface_hub import save_torch_state_dict

    save_torch_state_dict(state_dict=state, save_directory=save_dir, max_shard_size=shard_size, safe_serialization=True,
                          force_contiguous=True,
                          shared_tensors_to_discard=getattr(model, "_tied_weights_keys", None))


# ---- round-to-nearest quantisation of a skeleton (test / example helper)
def pseudo_quantize_tensor(w: torch.Tensor, w_bit: int = 4, group_size: int = 128):
    """Zero-point group quantisation, `awq/quantize/quantizer.py:74-109` (zero_point=True branch):
    returns (dequantised w, scales [out, in/g], zeros [out, in/g])."""
    org_shape = w.shape
    if group_size > 0:
        assert org_shape[-1] % group_size == 0
        w = w.reshape(-1, group_size)
    max_val = w.amax(dim=1, keepdim=True)
    min_val = w.amin(dim=1, keepdim=True)
    max_int = 2 ** w_bit - 1
    scales = (max_val - min_val).clamp(min=1e-5) / max_int
    zeros = (-torch.round(min_val / scales)).clamp_(0, max_int)
    w = (torch.clamp(torch.round(w / scales) + zeros, 0, max_int) - zeros) * scales
    return w.reshape(org_shape), scales.view(org_shape[0], -1), zeros.view(org_shape[0], -1)


def quantize_linears_rtn(model: nn.Module, quant_config: AwqConfig, layers=None) -> List[str]:
    """`_apply_quant` (`quantizer.py:228-262`) with the scale search skipped: pack every Linear of
    the decoder layers with plain round-to-nearest scales / zeros."""
    cls = _linear_class(quant_config.version)
    layers = find_decoder_layers(model) if layers is None else layers
    done = []
    with torch.no_grad():
        for layer in layers:
            named = exclude_layers_to_not_quantize(get_named_linears(layer), quant_config.modules_to_not_convert)
            for name, lin in named.items():
                lin = lin.half()
                g = quant_config.q_group_size if quant_config.q_group_size != -1 else lin.in_features
                lin.weight.data, scales, zeros = pseudo_quantize_tensor(lin.weight.data, quant_config.w_bit, g)
                if quant_config.version == "gemm":
                    scales, zeros = scales.t().contiguous(), zeros.t().contiguous()
                q = cls.from_linear(lin, quant_config.w_bit, quant_config.q_group_size, False, scales, zeros)
                set_op_by_name(layer, name, q)
                done.append(name)
    return done
