"""CPU oracle for the mlx-vlm generate hot path (Qwen2-VL first).

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a torch-CPU / numpy restatement of the reference algorithm
(/root/reference/mlx_vlm, v0.6.15) for the path named in BASELINE.json:
vision-tower patch embed + ViT attention prefill -> multimodal projector
(PatchMerger) -> KV-cache autoregressive LLM decode -> sampler.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker / reported CPU baseline.
The product package (`mlx-vlm_amd/`, import name `mlx_vlm_amd`) never imports
it and has no CPU fallback.

PARITY STATUS: pinned to the reference's own source, executed here; the arithmetic inside MLX's kernels
remains unpinned.
  * The reference computes through the un-vendored third-party runtime `mlx` (requirements.txt:1
    `mlx>=0.32.0`; uv.lock pins mlx 0.32.0, mlx-cpu 0.30.4).  `import mlx` fails in this container and there
    is no network; the reference's own tests hold no golden vectors for this path.
  * Pin 1 - the reference itself, run in this container (tests/golden/make_golden_ref.py ->
    tests/golden/qwen2_vl_tiny_ref.npz, checked by tests/test_oracle_ref_golden.py): the reference's files
    mlx_vlm/models/qwen2_vl/{config,vision,language,qwen2_vl}.py, models/{base,cache,rope_utils,mlp,
    activations}.py, sample_utils.py and generate/{ar,common}.py (generate_step) are imported UNMODIFIED from /root/reference and executed over
    `oracle/mlx_shim`, a torch-CPU stand-in that restates MLX's published op semantics (typed arrays, per-op
    rounding, weak-typed python scalars, fp32-accumulating matmul / fast ops).  Against those vectors the
    oracle is bit-exact in bf16 on the whole language-model path (prefill, KVCache decode, rope deltas), on the
    vision tower from the patch embeddings on, on the embedding merge, the rope-index tables (image, text-only,
    left-padded), the sampler filters and generate_step's tokens + bf16 logprobs on a text prompt; the patch-embed contraction agrees to 1 bf16 ulp (summation order).
    This pins the reference's GRAPH: op order, reshapes, dtype casts, rounding points, cache and position logic.
    The same method pins the nanoLLaVA path (oracle/llava_bunny.py; make_golden_ref_bunny.py ->
    llava_bunny_tiny_ref.npz; tests/test_oracle_ref_golden_bunny.py): models/llava_bunny/{config,vision,language,
    llava_bunny}.py incl. its ImageProcessor - bit-exact in bf16 from the patch embeddings on, image processor
    crc-exact, generate_step on a text prompt bit-exact, and the float32-promotion path the reference takes when
    handed float32 pixels (tokens identical, float32 log-probs to 2e-4).
  * Pin 2 - an independent implementation (tests/golden/make_golden.py -> qwen2_vl_tiny_hf.npz): HuggingFace
    transformers 5.15 `Qwen2VLForConditionalGeneration` in fp32 (same checkpoint format).  The reference-over-
    shim vectors agree with it to 1.3e-5, which validates the stand-in.
    The other families (llava_bunny, idefics2, phi3_v) have the same second pin against HF's SigLIP / Qwen2 /
    Idefics2 / CLIP / Phi-3 classes (tests/golden/make_golden_hf_families.py -> families_hf.npz, <= 2.2e-5).
  * Pin 3 - the reference's own M-RoPE contract (tests/test_rope_utils.py:366-407): fused Metal kernel ==
    pure-MLX path within atol 1e-4 in fp32.  The Metal kernel cannot run off-Metal (the reference itself takes
    the pure-MLX path there); the oracle carries both modes and is held to the same contract.
  * NOT pinned: accumulation order inside MLX's matmul / sdpa / conv kernels and the fused Metal rope kernel's
    last-ulp behaviour.  The HIP kernels are therefore compared with the tolerances written in the tests
    (bit-exact for integer / index work; bf16-ulp level for floating point).
"""
