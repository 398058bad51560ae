"""Prompt formatting for the built model families - host mirror of the reference's `mlx_vlm/prompt_utils.py`
(`MODEL_CONFIG` 33-117, `MessageFormatter` 262-552, `get_message_json` 555-591, `get_chat_template` 594-826,
`apply_chat_template` 829-995) for the message formats those families use:

    qwen2_vl, idefics2            content = [text item, image items ...]            (LIST_WITH_IMAGE)
    llava-qwen2 / bunny-llama     content = "<image>\\n" * n + prompt, one image     (IMAGE_TOKEN_NEWLINE, single-image models)
    phi3_v                        content = "<|image_1|><|image_2|>..." + prompt    (NUMBERED_IMAGE_TOKENS)
    anything else                 plain text messages

then the processor's (or its tokenizer's) chat template, or - when there is none - the reference's plain-prompt fallback
("User: ...\\nAssistant:").  Audio / video items and tool messages are outside the built path (the reference formats them
here too).  Pinned to the reference's own module, which needs no `mlx`: tests/golden/make_golden_prompts.py ->
prompts_ref.json, tests/test_prompt_utils_cpu.py."""
from __future__ import annotations

from typing import Any, Dict, List, Union

# model_type -> (kind, token).  kind: "list" = typed content list, "token" = marker text in front of the prompt,
# "numbered" = <|image_N|> markers
_FORMATS = {"qwen2_vl": ("list", None), "idefics2": ("list", None),
            "llava-qwen2": ("token", "<image>\n"), "bunny-llama": ("token", "<image>\n"),
            "phi3_v": ("numbered", None)}
_SINGLE_IMAGE_ONLY = {"llava-qwen2", "bunny-llama"}


def _text_of(content: Any) -> str:
    """text parts of an OpenAI-style content list (image / audio items are dropped: they arrive through `image=`)"""
    if isinstance(content, str):
        return content
    if isinstance(content, list):
        parts = [(it.get("text", "") or it.get("content", "")) for it in content
                 if isinstance(it, dict) and it.get("type", "") in ("text", "input_text")]
        parts = [p for p in parts if p]
        return " ".join(parts).strip() if parts else ""
    return str(content) if content else ""


def _role_content(item: Any):
    if isinstance(item, dict):
        return item.get("role", "user"), item.get("content")
    if hasattr(item, "role") and hasattr(item, "content"):
        return getattr(item, "role", "user"), getattr(item, "content", "")
    return None


def _is_tool_message(m: Dict[str, Any]) -> bool:
    return "tool_calls" in m or "tool_call_id" in m or m.get("role") == "tool"


def _tool_message(m: Dict[str, Any]) -> Dict[str, Any]:
    """reference 189-218: a copy whose function arguments are objects (JSON strings parsed), empty content instead of None"""
    import json

    out = dict(m)
    calls = out.get("tool_calls")
    if out.get("role") == "assistant" and calls and out.get("content") is None:
        out["content"] = ""
    if calls is None:
        return out
    fixed = []
    for c in calls:
        c = dict(c) if isinstance(c, dict) else c
        if isinstance(c, dict) and "function" in c:
            fn = dict(c["function"])
            if isinstance(fn.get("arguments", {}), str):
                try:
                    fn["arguments"] = json.loads(fn["arguments"])
                except (json.JSONDecodeError, TypeError):
                    fn["arguments"] = {}
            c["function"] = fn
        fixed.append(c)
    out["tool_calls"] = fixed
    return out


def get_message_json(model_name: str, prompt: str, role: str = "user", skip_image_token: bool = False,
                     skip_audio_token: bool = False, num_images: int = 0, num_audios: int = 0, **kwargs) -> Dict[str, Any]:
    """One chat message in the shape the model type's template expects (reference 555-591)."""
    name = model_name.lower()
    if name not in _FORMATS:
        raise ValueError(f"Unsupported model: {model_name}")
    if num_audios and not skip_audio_token and role == "user":
        raise NotImplementedError("audio inputs are outside the built hot path")
    if kwargs.get("video"):
        raise NotImplementedError("video inputs are outside the built hot path")
    if num_images > 1 and name in _SINGLE_IMAGE_ONLY:
        raise ValueError(f"Model {name} does not support multi-image chat. Please only use 1 image.")
    kind, token = _FORMATS[name]
    with_images = role == "user" and not skip_image_token and num_images > 0
    if kind == "list":
        content: List[Dict[str, str]] = [{"type": "text", "text": prompt, "content": prompt}]
        if with_images:
            content = content + [{"type": "image"}] * num_images
        return {"role": role, "content": content}
    if kind == "token":
        return {"role": role, "content": f"{token * num_images}{prompt}" if with_images else prompt}
    marks = "".join(f"<|image_{i + 1}|>" for i in range(num_images)) if with_images else ""
    return {"role": role, "content": f"{marks}{prompt}"}


def _plain_prompt(processor, messages, add_generation_prompt: bool) -> str:
    """the reference's fallback when no chat template exists (594-728): content lists flattened with the image token, a
    single user message returned as is, otherwise "Role: content" lines"""
    image_token = "<image>"
    for holder in (processor, getattr(processor, "tokenizer", None)):
        t = getattr(holder, "image_token", None)
        if isinstance(t, str) and t:
            image_token = t
            break

    def flat(content):
        if isinstance(content, str):
            return content
        if isinstance(content, list):
            parts = []
            for it in content:
                if isinstance(it, dict):
                    ty = it.get("type", "")
                    if ty in ("image", "image_url", "input_image"):
                        parts.append(image_token)
                    else:
                        text = it.get("text", "") or it.get("content", "")
                        if text:
                            parts.append(str(text))
                elif it is not None:
                    parts.append(str(it))
            out, prev_marker = [], False
            for p in parts:
                if not p:
                    continue
                marker = p in (image_token, "<video>", "<audio>")
                if prev_marker and not marker and not p[0].isspace():
                    out.append(" ")
                out.append(p)
                prev_marker = marker
            return "".join(out).strip()
        if isinstance(content, dict):
            return str(content.get("text", "") or content.get("content", "") or "")
        return str(content) if content is not None else ""

    norm = []
    for m in messages:
        if isinstance(m, str):
            norm.append({"role": "user", "content": m})
        elif isinstance(m, dict):
            norm.append({"role": m.get("role", "user"), "content": flat(m.get("content", ""))})
        else:
            norm.append({"role": "user", "content": str(m)})
    if not norm:
        return ""
    if len(norm) == 1 and norm[0]["role"] == "user":
        return norm[0]["content"]
    lines = []
    for m in norm:
        role, content = m.get("role", "user"), m.get("content", "")
        if role in ("system", "user", "assistant", "tool"):
            lines.append(f"{role.capitalize()}: {content}" if content else f"{role.capitalize()}:")
        else:
            lines.append(content if content else "")
    if add_generation_prompt:
        lines.append("Assistant:")
    return "\n".join(lines).strip()


def get_chat_template(processor, messages: List[Dict[str, Any]], add_generation_prompt: bool, tokenize: bool = False, **kwargs):
    """The processor's chat template (or its tokenizer's) over `messages`; the plain prompt when neither has one."""
    override = kwargs.get("chat_template", None)
    target = None
    if processor is not None and hasattr(processor, "apply_chat_template") and \
            (override is not None or getattr(processor, "chat_template", None) is not None):
        target = processor
    elif processor is not None and hasattr(getattr(processor, "tokenizer", None), "apply_chat_template") and \
            (override is not None or getattr(processor.tokenizer, "chat_template", None) is not None):
        target = processor.tokenizer
    if target is None:
        return _plain_prompt(processor, messages, add_generation_prompt)
    kw = dict(kwargs)
    if "enable_thinking" not in kw:           # as the reference: thinking off unless asked, where the template call can take it
        import inspect

        try:
            params = inspect.signature(target.apply_chat_template).parameters
            if "enable_thinking" in params or any(q.kind == inspect.Parameter.VAR_KEYWORD for q in params.values()):
                kw["enable_thinking"] = False
        except (TypeError, ValueError):
            pass
    try:
        return target.apply_chat_template(messages, tokenize=tokenize, add_generation_prompt=add_generation_prompt, **kw)
    except ValueError as e:
        if override is None and ("chat_template is not set" in str(e) or "no template argument was passed" in str(e)):
            return _plain_prompt(processor, messages, add_generation_prompt)
        raise
    except AttributeError:
        return _plain_prompt(processor, messages, add_generation_prompt)


def apply_chat_template(processor, config: Union[Dict[str, Any], Any], prompt: Union[str, Dict[str, Any], List[Any]],
                        add_generation_prompt: bool = True, return_messages: bool = False, num_images: int = 0,
                        num_audios: int = 0, **kwargs):
    """reference 829-995: a prompt string / one message / a conversation -> the formatted prompt string (or the messages).
    In a conversation, images stay with the user message that carries explicit image items; the rest go to the last user
    message."""
    cfg = config if isinstance(config, dict) else config.__dict__
    model_type = str(cfg["model_type"])
    if model_type.lower() not in _FORMATS:
        if isinstance(prompt, str):
            messages = [{"role": "user", "content": prompt}]
        elif isinstance(prompt, dict):
            messages = [dict(prompt, content=_text_of(prompt.get("content", "")))]
        elif isinstance(prompt, list):
            messages = []
            for it in prompt:
                if isinstance(it, str):
                    messages.append({"role": "user", "content": it})
                elif (rc := _role_content(it)) is not None:
                    msg = dict(it) if isinstance(it, dict) else {"role": rc[0]}
                    if rc[0] != "tool":
                        msg["content"] = _text_of(rc[1])
                    messages.append(msg)
        else:
            messages = [{"role": "user", "content": str(prompt)}]
        return messages if return_messages else get_chat_template(processor, messages, add_generation_prompt, **kwargs)

    messages: List[Dict[str, Any]] = []
    if isinstance(prompt, str):
        messages.append(get_message_json(model_type, prompt, num_images=num_images, num_audios=num_audios, **kwargs))
    elif isinstance(prompt, dict):
        if _is_tool_message(prompt):
            messages.append(_tool_message(prompt))
        else:
            messages.append(get_message_json(model_type, _text_of(prompt["content"]), prompt.get("role", "user"),
                                             num_images=num_images, num_audios=num_audios, **kwargs))
    elif isinstance(prompt, list):
        last_user, explicit = -1, [0] * len(prompt)
        for i, p in enumerate(prompt):
            if isinstance(p, str):
                last_user = i
            elif (rc := _role_content(p)) is not None and rc[0] not in ("system", "assistant", "tool"):
                last_user = i
                if isinstance(rc[1], list):
                    explicit[i] = sum(1 for it in rc[1] if isinstance(it, dict) and it.get("type") in ("image", "image_url", "input_image"))
        remaining, counts = num_images, []
        for c in explicit:
            c = min(c, remaining)
            counts.append(c)
            remaining -= c
        if remaining and last_user >= 0:
            counts[last_user] += remaining
        for i, p in enumerate(prompt):
            n_aud = num_audios if i == last_user else 0
            if isinstance(p, str):
                messages.append(get_message_json(model_type, p, skip_image_token=counts[i] == 0, skip_audio_token=n_aud == 0,
                                                 num_images=counts[i], num_audios=n_aud, **kwargs))
            elif isinstance(p, dict) and _is_tool_message(p):
                messages.append(_tool_message(p))                    # tool traffic goes to the template untouched
            elif (rc := _role_content(p)) is not None:
                role, content = rc
                quiet = role in ("system", "assistant")
                messages.append(get_message_json(model_type, _text_of(content), role, skip_image_token=counts[i] == 0 or quiet,
                                                 skip_audio_token=n_aud == 0 or quiet, num_images=counts[i], num_audios=n_aud,
                                                 **kwargs))
    if return_messages:
        return messages
    return get_chat_template(processor, messages, add_generation_prompt, **kwargs)
