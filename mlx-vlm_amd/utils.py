"""Loading / IO / prompt plumbing with the reference's surface for the hot path:

    load, load_model, load_config      reference mlx_vlm/utils.py:1065-1119, 736-987, 1175-1210
    get_model_and_args                 reference mlx_vlm/utils.py:588-635 (MODEL_REMAPPING 34-62)
    prepare_inputs                     reference mlx_vlm/utils.py:1918-2136 (single image + prompt)
    StoppingCriteria                   reference mlx_vlm/utils.py:2191-2249
    make_streaming_detokenizer         reference mlx_vlm/tokenizer_utils.py:406-410 (naive variant 19-86)

Host-only code: safetensors -> torch tensors -> Model.load_weights (device packing).
"""
from __future__ import annotations

import glob
import importlib
import json
import os
from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch

# model_type aliases of the reference (utils.py:34-62): the checkpoint's `model_type` -> package under models/.  The
# whole alias table is kept so that a lookup resolves exactly as it does there; families that are not built here then
# fail in get_model_and_args with the reference's own "Model type ... not supported." (utils.py:633-635).
# Built: qwen2_vl, llava_bunny (nanoLLaVA ships `model_type: "llava-qwen2"`, Bunny `"bunny-llama"`).
MODEL_REMAPPING: Dict[str, str] = {
    "llava-qwen2": "llava_bunny", "bunny-llama": "llava_bunny", "llava_qwen2": "fastvlm", "lfm2-vl": "lfm2_vl",
    "cohere2_vision": "aya_vision", "jvlm": "jina_vlm", "phi4-siglip": "phi4_siglip", "sam3_video": "sam3",
    "sam3.1_video": "sam3_1", "granite-vision": "granite_vision", "granite4-vision": "granite4_vision",
    "granite4_vision": "granite4_vision", "rf-detr": "rfdetr", "falcon-perception": "falcon_perception",
    "nemotronh_nano_omni_reasoning_v3": "nemotron_h_nano_omni", "cohere2moe": "cohere2_moe",
    "unlimited-ocr": "unlimited_ocr", "mistral": "llama", "phi-msft": "phixtral", "falcon_mamba": "mamba",
    "joyai_llm_flash": "deepseek_v3", "kimi_k2": "deepseek_v3", "minimax_m2": "minimax", "iquestcoder": "llama",
    "nemotron-nas": "nemotron_nas", "inkling_mm_model": "inkling", "lille-130m": "lille_130m",
}


def get_model_and_args(config: dict):
    """reference utils.py:588-635"""
    model_type = config["model_type"].lower()
    model_type = MODEL_REMAPPING.get(model_type, model_type)
    try:
        arch = importlib.import_module(f"{__package__}.models.{model_type}")
    except ImportError as e:
        raise ValueError(f"Model type {model_type} not supported.") from e
    return arch, model_type


def load_config(model_path: str) -> dict:
    """reference utils.py:1175-1210"""
    p = os.path.join(model_path, "config.json")
    if not os.path.exists(p):
        raise FileNotFoundError(f"Config not found at {model_path}")
    with open(p) as f:
        config = json.load(f)
    g = os.path.join(model_path, "generation_config.json")
    if os.path.exists(g):
        try:
            with open(g) as f:
                gen = json.load(f)
            if "eos_token_id" in gen:
                config["eos_token_id"] = gen["eos_token_id"]
        except Exception:
            pass
    return config


def read_sanitized_weights(model_path: str, model, config: dict) -> Dict[str, torch.Tensor]:
    """The checkpoint of `model_path` as host tensors under the names `model.load_weights` takes (reference
    utils.py:781-801 file discovery, 846-870 sanitize of the model and of its vision tower)."""
    from safetensors.torch import load_file

    files = sorted(glob.glob(os.path.join(model_path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"No safetensors found in {model_path}")
    weights: Dict[str, torch.Tensor] = {}
    for f in files:
        weights.update(load_file(f))
    from .models import quantized as Qz

    Qz.check_quantization(config.get("quantization"))          # affine 4-bit / group 64 (utils.py:916-967)
    weights = model.sanitize(weights)
    vt = {k: v for k, v in weights.items() if k.startswith("vision_tower.")}
    vt = {"vision_tower." + k: v for k, v in model.vision_tower.sanitize(
        {k[len("vision_tower."):]: v for k, v in vt.items()}).items()}
    return {**{k: v for k, v in weights.items() if not k.startswith("vision_tower.")}, **vt}


def load_model(model_path: str, lazy: bool = False, device="cuda", **kwargs):
    """reference utils.py:736-987: config -> model class -> sanitize -> load_weights."""
    config = load_config(model_path)
    if not glob.glob(os.path.join(model_path, "*.safetensors")):
        raise FileNotFoundError(f"No safetensors found in {model_path}")
    arch, _ = get_model_and_args(config)
    mc = arch.ModelConfig.from_dict(config)
    model = arch.Model(mc, device=device, **kwargs)
    model.load_weights(read_sanitized_weights(model_path, model, config))
    return model


def freeze_heap():
    """Take everything allocated so far (torch, transformers, the model objects: ~4e5 GC-tracked containers) out of the
    cyclic garbage collector's view.  A generation-2 pass over that heap costs 90-150 ms of host time wherever it
    happens to trigger - in the middle of a prefill enqueue or a decode loop - and finds nothing: those objects live
    as long as the process.  Reference-counted frees are unaffected.  VLM_GC_FREEZE=0 opts out."""
    import gc

    if os.environ.get("VLM_GC_FREEZE", "1") != "0":
        gc.collect()
        gc.freeze()


def cpu_quota() -> int:
    """CPUs this process may actually use: the cgroup v2 / v1 CFS quota when there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(period)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = int(f.read()), int(g.read())
            if q > 0:
                n = min(n, max(1, q // period))
        except (OSError, ValueError):
            pass
    return n


def host_threads_per_rank() -> int:
    """This rank's share of the CPU quota: the ranks of one node (LOCAL_WORLD_SIZE, set by torchrun; WORLD_SIZE as the
    single-node fallback) share ONE cgroup budget, so eight ranks each sizing a pool for the whole quota is the CFS-throttle
    stall of profiles/r02_continuous_diag.txt times eight."""
    try:
        local_ws = int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1)
    except ValueError:
        local_ws = 1
    return max(1, cpu_quota() // max(1, local_ws))


def fit_host_threads():
    """torch sizes its CPU thread pool from the cores it SEES (256 on an MI355X host); inside a container with a CFS quota
    (16 cores on the boxes measured) any CPU tensor op then wakes ~128 OpenMP workers that keep spinning after the region,
    the cgroup's budget is gone within milliseconds and the whole process - the decode loop included - is frozen for the
    rest of the 100 ms period (profiles/r02_continuous_diag.txt: 80-90 ms stalls in a 2.7 MB copy).  The engine's own host
    path uses no parallel CPU ops; this caps torch's pool at this rank's share of the quota (quota // LOCAL_WORLD_SIZE, at
    least 1) for whatever else runs in the process.  VLM_FIT_HOST_THREADS=0 opts out."""
    if os.environ.get("VLM_FIT_HOST_THREADS", "1") == "0":
        return
    q = host_threads_per_rank()
    if torch.get_num_threads() > q:
        torch.set_num_threads(q)


def load(path_or_hf_repo: str, adapter_path=None, lazy: bool = False, revision=None, strict: bool = True, **kwargs):
    """reference utils.py:1065-1119 -> (model, processor).  Local paths only (no network in this build)."""
    if adapter_path is not None:
        raise NotImplementedError("LoRA adapters are out of scope (SURVEY §2.1 trainer)")
    if not os.path.isdir(path_or_hf_repo):
        raise FileNotFoundError(f"{path_or_hf_repo} is not a local model directory")
    model = load_model(path_or_hf_repo, lazy=lazy, **kwargs)
    processor = load_processor(path_or_hf_repo, model.config)
    freeze_heap()
    fit_host_threads()
    return model, processor


def load_processor(model_path: str, config):
    """reference utils.py:1243-1276: processor with .tokenizer, .detokenizer and tokenizer.stopping_criteria."""
    from transformers import AutoTokenizer

    from .models.qwen2_vl.processing_qwen2_vl import Qwen2VLImageProcessor, Qwen2VLProcessor

    tok = AutoTokenizer.from_pretrained(model_path)
    ip_kwargs = {}
    pp = os.path.join(model_path, "preprocessor_config.json")
    pc = {}
    if os.path.exists(pp):
        with open(pp) as f:
            pc = json.load(f)
    mt = str(getattr(config, "model_type", "") or "").lower()
    if MODEL_REMAPPING.get(mt, mt) == "llava_bunny":
        # reference utils.py:1260-1270: models with a BaseImageProcessor get the bare tokenizer with the image
        # processor attached; prepare_inputs splits the prompt at "<image>" (utils.py:2064-2095)
        from .models.llava_bunny import ImageProcessor

        tok.image_processor = ImageProcessor(**{k: pc[k] for k in ("image_mean", "image_std") if k in pc})
        tok.image_token_index = getattr(config, "image_token_index", -200)
        tok.tokenizer = tok
        eos = config.eos_token_id if getattr(config, "eos_token_id", None) is not None else tok.eos_token_id
        tok.stopping_criteria = StoppingCriteria(eos if isinstance(eos, list) else [eos], tok)
        tok.detokenizer = _pick_detokenizer(model_path, tok)
        return tok
    if MODEL_REMAPPING.get(mt, mt) == "idefics2":
        # reference idefics2/processing_idefics2.py:165-214 (Idefics2Processor.from_pretrained): tokenizer + transformers' image
        # processor configuration (size / mean / std / do_image_splitting) + image_seq_len of processor_config.json
        from .models.idefics2 import Idefics2ImageProcessor, Idefics2Processor

        ipk = {k: pc[k] for k in ("size", "image_mean", "image_std", "rescale_factor", "do_image_splitting") if k in pc}
        pk = {}
        ppath = os.path.join(model_path, "processor_config.json")
        if os.path.exists(ppath):
            with open(ppath) as f:
                pcfg = json.load(f)
            if "image_seq_len" in pcfg:
                pk["image_seq_len"] = pcfg["image_seq_len"]
            ipk.update({k: v for k, v in (pcfg.get("image_processor") or {}).items() if k in ("size",)})
        proc = Idefics2Processor(Idefics2ImageProcessor(**ipk), tok, chat_template=getattr(tok, "chat_template", None), **pk)
        eos = config.eos_token_id if getattr(config, "eos_token_id", None) is not None else tok.eos_token_id
        tok.stopping_criteria = StoppingCriteria(eos if isinstance(eos, list) else [eos], tok)
        proc.detokenizer = _pick_detokenizer(model_path, tok)
        return proc
    if MODEL_REMAPPING.get(mt, mt) == "phi3_v":
        # reference phi3_v/processing_phi3_v.py:596-660 (Phi3VProcessor.from_pretrained): tokenizer + the HD image processor
        # with num_crops / num_img_tokens / mean / std of preprocessor_config.json, chat template from the tokenizer
        from .models.phi3_v import Phi3VImageProcessor, Phi3VProcessor

        ipk = {k: pc[k] for k in ("num_crops", "num_img_tokens", "image_mean", "image_std") if k in pc}
        proc = Phi3VProcessor(Phi3VImageProcessor(**ipk), tok, chat_template=getattr(tok, "chat_template", None))
        eos = config.eos_token_id if getattr(config, "eos_token_id", None) is not None else tok.eos_token_id
        tok.stopping_criteria = StoppingCriteria(eos if isinstance(eos, list) else [eos], tok)
        proc.detokenizer = _pick_detokenizer(model_path, tok)
        return proc
    if pc:
        for k in ("image_mean", "image_std", "min_pixels", "max_pixels", "patch_size", "temporal_patch_size", "merge_size"):
            if k in pc:
                ip_kwargs[k] = pc[k]
    proc = Qwen2VLProcessor(Qwen2VLImageProcessor(**ip_kwargs), tok)
    eos = config.eos_token_id if getattr(config, "eos_token_id", None) is not None else tok.eos_token_id
    tok.stopping_criteria = StoppingCriteria(eos if isinstance(eos, list) else [eos], tok)
    proc.detokenizer = _pick_detokenizer(model_path, tok)
    return proc


def _pick_detokenizer(model_path, tok):
    """reference tokenizer_utils.py:453-480: the streaming detokenizer that fits tokenizer.json's decoder"""
    from .tokenizer_utils import detokenizer_class_for

    return detokenizer_class_for(model_path)(tok)


class StoppingCriteria:
    """reference utils.py:2191-2249"""

    def __init__(self, eos_token_ids: List[int], tokenizer=None, additional_eos_token_ids: Optional[List[int]] = None):
        self.tokenizer = tokenizer
        self.additional_eos_token_ids = list(dict.fromkeys(additional_eos_token_ids or ()))
        self.reset(eos_token_ids)

    def add_eos_token_ids(self, new_eos_token_ids: Union[int, str, List[Union[int, str]], None] = None):
        if new_eos_token_ids is None:
            return
        if self.tokenizer is None:
            raise ValueError("Processor is not provided")
        if isinstance(new_eos_token_ids, (str, int)):
            new_eos_token_ids = [new_eos_token_ids]
        resolved = []
        for token in new_eos_token_ids:
            if isinstance(token, int):
                resolved.append(token)
            elif isinstance(token, str):
                resolved.append(self.tokenizer.encode(" " + token, add_special_tokens=False)[-1])
        self.eos_token_ids.extend(resolved)

    def reset(self, eos_token_ids: Optional[List[int]] = None):
        eos_token_ids = eos_token_ids if eos_token_ids is not None else self.tokenizer.eos_token_ids
        if isinstance(eos_token_ids, int):
            eos_token_ids = [eos_token_ids]
        resolved = list(eos_token_ids)
        resolved.extend(t for t in self.additional_eos_token_ids if t not in resolved)
        if getattr(self, "eos_token_ids", None) != resolved:
            self.eos_token_ids = resolved

    def __call__(self, input_ids) -> bool:
        return input_ids in self.eos_token_ids


class ThinkingBudgetCriteria:
    """Budget on the tokens generated inside a thinking block (reference utils.py:2252-2335, same constructor, attributes
    and call protocol).  The generation loop shows it every token (`criteria(token)`, dispatch.py:1016-1018); once more than
    `thinking_budget` tokens have been produced inside an open block it hands out, one per call, the ids of "\\n" + the end
    marker, and generate_step feeds the pending id INSTEAD of the sampled token (`pop_forced_token_id`, ar.py:510-513)."""

    def __init__(self, tokenizer, thinking_budget: int, thinking_end_token: str = "</think>",
                 thinking_start_token: Optional[str] = None, enable_thinking: bool = False,
                 prompt_preopens_thinking: bool = False):
        last_id = lambda text: tokenizer.encode(text, add_special_tokens=False)[-1]   # noqa: E731
        self.tokenizer, self.thinking_budget = tokenizer, thinking_budget
        self.enable_thinking, self.prompt_preopens_thinking = enable_thinking, prompt_preopens_thinking
        self.thinking_end_token_id = last_id(thinking_end_token)
        self.thinking_start_token_id = last_id(thinking_start_token)
        nl = tokenizer.encode("\n", add_special_tokens=False)
        self._forced_sequence: List[int] = ([nl[-1]] if nl else []) + [self.thinking_end_token_id]
        self.forced_token_id = None
        self.reset_thinking_state()

    def reset_thinking_state(self):
        """between generations: a prompt that already opened the block starts inside it"""
        self.in_thinking = bool(self.enable_thinking and self.prompt_preopens_thinking)
        self.thinking_token_count, self.budget_exceeded, self._forced_index = 0, False, 0

    def __call__(self, token_id: int) -> Optional[int]:
        if token_id == self.thinking_start_token_id and self.enable_thinking:
            self.in_thinking = True
            return None
        if token_id == self.thinking_end_token_id:          # the block closed (by the model or by the forced sequence)
            self.in_thinking, self.budget_exceeded, self._forced_index = False, False, 0
            return None
        if self.in_thinking:
            self.thinking_token_count += 1
            self.budget_exceeded = self.budget_exceeded or self.thinking_token_count > self.thinking_budget
        pending = None
        if self.budget_exceeded and self._forced_index < len(self._forced_sequence):
            pending = self._forced_sequence[self._forced_index]
            self._forced_index += 1
        self.forced_token_id = pending
        return pending

    def pop_forced_token_id(self) -> Optional[int]:
        if not self.enable_thinking or self.forced_token_id is None:
            return None
        forced, self.forced_token_id = self.forced_token_id, None
        return forced


class NaiveStreamingDetokenizer:
    """reference tokenizer_utils.py:19-86 (NaiveStreamingDetokenizer): decode the running token list,
    emit the new text once it no longer ends in an incomplete UTF-8 sequence."""

    def __init__(self, tokenizer):
        self._tokenizer = tokenizer
        self.reset()

    def reset(self):
        self.offset = 0
        self.tokens: List[int] = []
        self._text = ""
        self._current_tokens: List[int] = []
        self._current_text = ""

    def add_token(self, token, skip_special_token_ids=()):
        if token in skip_special_token_ids:
            return
        self._current_tokens.append(token)
        self.tokens.append(token)

    def finalize(self):
        self._text += self._tokenizer.decode(self._current_tokens)
        self._current_tokens = []
        self._current_text = ""

    @property
    def text(self):
        if self._current_tokens:
            self._current_text = self._tokenizer.decode(self._current_tokens)
            if self._current_text.endswith("�"):
                self._current_text = self._current_text[:-1]
        if self._current_text and self._current_text[-1] == "\n":
            self._text += self._current_text
            self._current_tokens.clear()
            self._current_text = ""
        return self._text + self._current_text

    @property
    def last_segment(self):
        text = self.text
        seg = text[self.offset:]
        self.offset = len(text)
        return seg


def make_streaming_detokenizer(processor):
    """An isolated, reset detokenizer for one generation (reference tokenizer_utils.py:406-410): a copy of the one
    `load_processor` picked for the tokenizer (SPM / byte-level BPE / naive), or a naive one for a bare tokenizer."""
    import copy

    det = getattr(processor, "detokenizer", None)
    if det is not None:
        det = copy.copy(det)
        det.reset()
        return det
    tok = processor.tokenizer if hasattr(processor, "tokenizer") else processor
    if not hasattr(tok, "decode"):
        return None
    return NaiveStreamingDetokenizer(tok)


def prepare_inputs(processor, images=None, prompts=None, **kwargs) -> Dict[str, Any]:
    """reference utils.py:1918-2136, single-sequence form: -> input_ids [1, L] (np.int64), attention_mask,
    pixel_values [N, C*T*ps*ps] f32, image_grid_thw [n_img, 3]."""
    if images is not None and not isinstance(images, (list, tuple)):
        images = [images]
    if hasattr(processor, "image_processor") and hasattr(processor, "image_token_index"):
        # BaseImageProcessor models (nanoLLaVA), reference utils.py:2064-2095
        from .models.llava_bunny.processing import assemble_input_ids
        from .models.qwen2_vl.processing_qwen2_vl import load_image

        plist = [prompts] if isinstance(prompts, str) else list(prompts)
        if processor.pad_token is None:
            processor.pad_token = processor.eos_token
        if not images:
            enc = processor(plist, padding=True)
            return {"input_ids": np.asarray(enc["input_ids"], dtype=np.int64),
                    "attention_mask": np.asarray(enc["attention_mask"], dtype=np.int32)}
        ids, mask = assemble_input_ids(lambda chunk: processor(chunk).input_ids, plist, processor.pad_token_id,
                                       processor.image_token_index)
        pix = processor.image_processor.preprocess([load_image(im) for im in images])
        return {"input_ids": ids, "pixel_values": np.stack(pix), "attention_mask": mask}
    imgs = None
    if images:
        from .models.qwen2_vl.processing_qwen2_vl import load_image

        imgs = [load_image(im) for im in images]
    out = processor(images=imgs, text=prompts if isinstance(prompts, str) else list(prompts))
    return {k: v for k, v in out.items()}
