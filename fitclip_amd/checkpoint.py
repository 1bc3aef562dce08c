"""On-disk formats either side of the encoder: Lightning `.ckpt` files and bare OpenAI-CLIP-named state dicts.

Mirrors (behaviour, not code):
  * `util/checkpoint_utils.py:9-12`  -> `state_dict_from_checkpoint_path`: `checkpoint["state_dict"]` filtered by a key
    prefix (a missing trailing dot is added), prefix stripped;
  * `scripts/checkpoint_to_state_dict.py` -> `python -m fitclip_amd.checkpoint INPUT [--prefix encoder.model.] > out.pt`;
  * `scripts/prepare_trained_clip_checkpoint_for_evaluation.py`, `scripts/prepare_trained_checkpoint_for_evaluation.py`,
    `scripts/open_clip_checkpoint_to_model.py`, `scripts/apply_wise_ft.py` -> the sub-commands `prepare-clip`, `prepare`,
    `open-clip`, `apply-wise-ft` (file in, file out);
  * `aligner/text_video_retrieval.py:101-131` -> `load_module_state_dict`: a plain retrieval module silently drops the
    `teacher*` keys of a teacher-student checkpoint, and reports the other mismatches with torch's own wording;
  * the module-level keys a Lightning checkpoint of the reference holds, in `nn.Module.state_dict()` order (a module's own
    parameters first, then its children in registration order): `logit_scale`, `max_logit_scale` (both `nn.Parameter`s of
    shape [1], `aligner/video_text_module.py:32-34`), for the distillation module `teacher_student_logit_scale`
    (`aligner/teacher_student.py:70`), then `encoder.<param>`, `teacher.<param>` and, when the module was built with
    `prompts`, `tokenized_prompts.input_ids` / `teacher_tokenized_prompts.input_ids` (`teacher_student.py:86-91`);
  * `reference_parameter_order`: the index every parameter has in `torch.optim.AdamW(self.parameters())`
    (`aligner/cli.py:129-132`), which is how `optimizer_states[0]["state"]` of a reference checkpoint is keyed.

Only LOCAL paths are read (there is no egress; the reference's `cached_path(url)` fetch has no counterpart).  A path
may be a pipe (process substitution), as in the reference's README recipes: it is drained into memory first because
`torch.load` needs a seekable file (`clip_video_text_encoder.py:35-41` copies it to a temp file for the same reason).
"""
from __future__ import annotations

import argparse
import io
import math
import os
import sys
from collections import OrderedDict
from typing import Any, Dict, List, Mapping, MutableMapping, NamedTuple, Union

import torch

TYPE_PATH = Union[str, "os.PathLike[str]"]


def _torch_load(file: Any, trusted: bool) -> Any:
    """`weights_only=True` first: bare state dicts and most Lightning checkpoints are tensors + plain containers and
    never need the unpickler.  Only a file the CALLER declares trusted (`trusted=True` / FITCLIP_TRUST_CHECKPOINTS=1) is
    retried with the full unpickler, which can execute code embedded in the file (Lightning checkpoints that pickle
    hyper-parameter objects need it - the reference's `torch.load` did this unconditionally)."""
    import pickle
    try:
        return torch.load(file, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not (trusted or os.environ.get("FITCLIP_TRUST_CHECKPOINTS") == "1"):
            raise pickle.UnpicklingError(
                f"{e}\nThis checkpoint holds pickled objects besides tensors.  If you trust its origin, load it with "
                "trusted=True or set FITCLIP_TRUST_CHECKPOINTS=1.") from None
        if hasattr(file, "seek"):
            file.seek(0)
        return torch.load(file, map_location="cpu", weights_only=False)


def _load(path: TYPE_PATH, trusted: bool = False) -> Any:
    path = os.fspath(path)
    if "://" in path:
        raise FileNotFoundError(f"{path!r}: remote checkpoints cannot be fetched here (no network); pass a local file")
    if os.path.exists(path) and not os.path.isdir(path) and not os.path.isfile(path):  # a pipe
        with open(path, "rb") as f:
            return _torch_load(io.BytesIO(f.read()), trusted)
    return _torch_load(path, trusted)


def load_state_dict_file(path: TYPE_PATH) -> MutableMapping[str, torch.Tensor]:
    """A bare state dict, or the `state_dict` of a Lightning checkpoint with the `encoder.model.` prefix stripped."""
    obj = _load(path)
    if isinstance(obj, Mapping) and "state_dict" in obj and isinstance(obj["state_dict"], Mapping):
        return strip_prefix(obj["state_dict"], "encoder.model.")
    return obj


def strip_prefix(state_dict: Mapping[str, torch.Tensor], prefix: str = "") -> MutableMapping[str, torch.Tensor]:
    prefix += "" if prefix.endswith(".") or not prefix else "."
    return {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}


def state_dict_from_checkpoint_path(checkpoint_path: TYPE_PATH, prefix: str = "") -> MutableMapping[str, torch.Tensor]:
    return strip_prefix(_load(checkpoint_path)["state_dict"], prefix)


def trainable_parameter_indices(module: Any, fit_temperature: bool = True) -> "OrderedDict[int, str]":
    """index -> key of the parameters that receive gradients (and therefore own an entry in the optimiser's `state`):
    the two temperatures when they are fitted (video_text_module.py:32, teacher_student.py:70-71) and every encoder
    parameter; the teacher is frozen (teacher_student.py:75-76)."""
    out: "OrderedDict[int, str]" = OrderedDict()
    for i, key in enumerate(reference_parameter_order(module)):
        if key in ("logit_scale", "teacher_student_logit_scale"):
            if fit_temperature:
                out[i] = key
        elif key.startswith("encoder."):
            out[i] = key
    return out


# ------------------------------------------------------------------------------------------- module <-> state dict
class IncompatibleKeys(NamedTuple):
    missing_keys: List[str]
    unexpected_keys: List[str]


def _scalar(v: float) -> torch.Tensor:
    return torch.tensor([float(v)])  # the reference's temperatures are nn.Parameters of shape [1]


_PROMPT_KEYS = ("tokenized_prompts", "teacher_tokenized_prompts")  # nn.ParameterDicts of {"input_ids": [n, L]}


def module_state_dict(module: Any) -> "OrderedDict[str, torch.Tensor]":
    """The keys, shapes and ORDER a Lightning checkpoint of the reference's module holds for the same object graph."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    sd["logit_scale"] = _scalar(module.logit_scale)
    sd["max_logit_scale"] = _scalar(module.max_logit_scale)
    teacher = getattr(module, "teacher", None)
    if teacher is not None:
        sd["teacher_student_logit_scale"] = _scalar(module.teacher_student_logit_scale)
    for k, v in module.encoder.state_dict().items():
        sd[f"encoder.{k}"] = v
    if teacher is not None:
        for k, v in teacher.state_dict().items():
            sd[f"teacher.{k}"] = v
        for attr in _PROMPT_KEYS:
            ids = getattr(module, attr, None)
            if ids is not None:
                sd[f"{attr}.input_ids"] = ids
    return sd


def reference_parameter_order(module: Any) -> List[str]:
    """State-dict keys of everything `self.parameters()` yields on the reference's module, in that order - the order
    `torch.optim.AdamW(self.parameters())` numbers them (cli.py:129-132): the module's own parameters (`logit_scale`,
    `max_logit_scale`, `teacher_student_logit_scale`), then `encoder.*`, `teacher.*` and the prompt ids.  Frozen ones
    (max_logit_scale, the teacher, the prompts; the temperatures with fit_temperature=False) are numbered too but
    never get optimiser state."""
    names = ["logit_scale", "max_logit_scale"]
    teacher = getattr(module, "teacher", None)
    if teacher is not None:
        names.append("teacher_student_logit_scale")
    names += [f"encoder.{k}" for k, _ in module.encoder.named_parameters()]
    if teacher is not None:
        names += [f"teacher.{k}" for k, _ in teacher.named_parameters()]
        names += [f"{attr}.input_ids" for attr in _PROMPT_KEYS if getattr(module, attr, None) is not None]
    return names


def save_checkpoint(module: Any, path: TYPE_PATH, **extra: Any) -> None:
    """Writes `{"state_dict": ...}` (the part of a Lightning checkpoint the reference's tools read)."""
    torch.save({"state_dict": module_state_dict(module), **extra}, os.fspath(path))


def load_module_state_dict(module: Any, state_dict: Mapping[str, torch.Tensor], strict: bool = True) -> IncompatibleKeys:
    """`load_state_dict` of the module family in `fitclip_amd.retrieval`.

    A module WITHOUT a teacher ignores every key that starts with "teacher" (so a distillation checkpoint can be
    evaluated with the plain retrieval module, text_video_retrieval.py:104-111); everything else follows
    `nn.Module.load_state_dict`: with `strict`, missing / unexpected keys raise a RuntimeError with torch's wording.
    """
    own = module_state_dict(module)
    has_teacher = getattr(module, "teacher", None) is not None
    unexpected = [k for k in state_dict if k not in own and (has_teacher or not k.startswith("teacher"))]
    # `max_logit_scale` is a constant the reference happens to store as a parameter: files written before this key
    # was emitted (and bare `{"logit_scale", "encoder.*"}` dictionaries) stay loadable
    missing = [k for k in own if k not in state_dict and k != "max_logit_scale"]
    if strict and (unexpected or missing):
        msgs = []
        if unexpected:
            msgs.append("Unexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected) + ". ")
        if missing:
            msgs.append("Missing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing) + ". ")
        raise RuntimeError(f"Error(s) in loading state_dict for {type(module).__name__}:\n\t" + "\n\t".join(msgs))

    def sub(prefix: str) -> Dict[str, torch.Tensor]:
        return {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix) and k in own}

    if "logit_scale" in state_dict:
        module.logit_scale = float(state_dict["logit_scale"])   # shape [1] in a reference file, 0-d accepted too
    if "max_logit_scale" in state_dict:
        module.max_logit_scale = float(state_dict["max_logit_scale"])
    enc = sub("encoder.")
    if enc:
        module.encoder.load_state_dict(enc, strict=False)
    if has_teacher:
        tch = sub("teacher.")
        if tch:
            module.teacher.load_state_dict(tch, strict=False)
        if "teacher_student_logit_scale" in state_dict:
            module.teacher_student_logit_scale = float(state_dict["teacher_student_logit_scale"])
        for attr in _PROMPT_KEYS:
            key = f"{attr}.input_ids"
            if key in state_dict and key in own:
                setattr(module, attr, state_dict[key].to(getattr(module, attr).device).long())
    assert not math.isnan(module.logit_scale), "a checkpoint's module-level logit_scale is never NaN"
    return IncompatibleKeys(missing, unexpected)


def load_checkpoint(module: Any, path: TYPE_PATH, strict: bool = True) -> IncompatibleKeys:
    return load_module_state_dict(module, _load(path)["state_dict"], strict=strict)


# --------------------------------------------------------------------------------------------- file-level tools
def prepare_trained_clip_checkpoint(input_path: TYPE_PATH, output_path: TYPE_PATH, prefix: str = "encoder.model.") -> None:
    """Lightning checkpoint -> bare CLIP state dict FILE that `load_clip_model` / `clip.load` accept: prefix stripped and
    the `logit_scale` the training module dropped re-created as NaN
    (`scripts/prepare_trained_clip_checkpoint_for_evaluation.py`)."""
    state_dict = state_dict_from_checkpoint_path(input_path, prefix=prefix)
    state_dict["logit_scale"] = torch.tensor(float("nan"))
    torch.save(state_dict, os.fspath(output_path))


def prepare_trained_checkpoint(input_path: TYPE_PATH, output_path: TYPE_PATH, prefix: str = "encoder.model.") -> None:
    """Keeps the checkpoint dictionary (epoch, optimizer state, ...) but replaces its `state_dict` by the prefix-stripped
    one (`scripts/prepare_trained_checkpoint_for_evaluation.py`)."""
    checkpoint = _load(input_path)
    checkpoint["state_dict"] = strip_prefix(checkpoint["state_dict"], prefix)
    torch.save(checkpoint, os.fspath(output_path))


def open_clip_checkpoint_to_model(input_path: TYPE_PATH, output_path: TYPE_PATH) -> None:
    """open_clip training checkpoint -> bare state dict (`scripts/open_clip_checkpoint_to_model.py`): the wrapper prefix
    (`model.` or `module.`, decided by the FIRST key) is cut from the front of every key."""
    state_dict = _load(input_path)["state_dict"]
    first_key = next(iter(state_dict))
    prefix = next(p for p in ("model", "module") if first_key.startswith(p + "."))  # StopIteration if neither: as there
    torch.save({k[len(prefix) + 1:]: v for k, v in state_dict.items()}, os.fspath(output_path))


def apply_wise_ft(input_path1: TYPE_PATH, input_path2: TYPE_PATH, output_path: TYPE_PATH, weight_for_2: float = 0.5,
                  device: Union[str, torch.device] = "cuda") -> None:
    """WiSE-FT of two CLIP checkpoint FILES, written as a bare state dict (`scripts/apply_wise_ft.py`): both models are
    loaded, a NaN `logit_scale` is re-created where the file lacks one, the parameters are blended on the device
    (`fc_wise`, bit-identical to `(1 - w) * p1 + w * p2`)."""
    from .clip_model import load_clip_model
    from .wise import wise_state_dict
    model1 = load_clip_model(os.fspath(input_path1), precision="fp32", device=device)
    model2 = load_clip_model(os.fspath(input_path2), precision="fp32", device=device)
    blended = wise_state_dict(model1, model2, weight_for_2=weight_for_2)
    torch.save({k: v.cpu() for k, v in blended.items()}, os.fspath(output_path))


# ----------------------------------------------------------------------------------------------------------- CLI
def main(argv: Union[List[str], None] = None) -> None:
    """`python -m fitclip_amd.checkpoint INPUT [--prefix P]`  (state dict to stdout, as checkpoint_to_state_dict.py), or
    one of the sub-commands `prepare-clip`, `prepare`, `open-clip`, `apply-wise-ft`."""
    argv = list(sys.argv[1:] if argv is None else argv)
    tools = {"prepare-clip": prepare_trained_clip_checkpoint, "prepare": prepare_trained_checkpoint}
    if argv and argv[0] in tools:
        parser = argparse.ArgumentParser(prog=f"fitclip_amd.checkpoint {argv[0]}")
        parser.add_argument("input_path", metavar="INPUT_FILE")
        parser.add_argument("output_path", metavar="OUTPUT_FILE")
        parser.add_argument("--prefix", default="encoder.model.")
        args = parser.parse_args(argv[1:])
        tools[argv[0]](args.input_path, args.output_path, prefix=args.prefix)
        return
    if argv and argv[0] == "open-clip":
        parser = argparse.ArgumentParser(prog="fitclip_amd.checkpoint open-clip")
        parser.add_argument("input_path", metavar="INPUT_FILE")
        parser.add_argument("output_path", metavar="OUTPUT_FILE")
        args = parser.parse_args(argv[1:])
        open_clip_checkpoint_to_model(args.input_path, args.output_path)
        return
    if argv and argv[0] == "apply-wise-ft":
        parser = argparse.ArgumentParser(prog="fitclip_amd.checkpoint apply-wise-ft",
                                         description="Weight-space ensemble (WiSE-FT, arXiv 2109.01903) of two CLIP "
                                                     "checkpoints.")
        parser.add_argument("input_path1", metavar="INPUT_FILE_1")
        parser.add_argument("input_path2", metavar="INPUT_FILE_2")
        parser.add_argument("output_path", metavar="OUTPUT_FILE")
        parser.add_argument("--weight-for-2", type=float, default=0.5)
        args = parser.parse_args(argv[1:])
        apply_wise_ft(args.input_path1, args.input_path2, args.output_path, weight_for_2=args.weight_for_2)
        return
    parser = argparse.ArgumentParser(description="Lightning checkpoint -> bare (prefix-stripped) state dict on stdout")
    parser.add_argument("input_path", metavar="INPUT_FILE")
    parser.add_argument("--prefix", default="encoder.model.")
    args = parser.parse_args(argv)
    torch.save(state_dict_from_checkpoint_path(args.input_path, prefix=args.prefix), sys.stdout.buffer)


if __name__ == "__main__":
    main()
